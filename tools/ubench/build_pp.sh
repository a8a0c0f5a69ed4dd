#!/bin/bash
# builds tools/ubench/build/pp_<tag> for every "tag[=flags]" argument in parallel, e.g.
#   bash tools/ubench/build_pp.sh plain lab=-DAE_GEMM_LAB abl1=-DAE_PP_LAB=1
cd "$(dirname "$0")/build" || exit 1
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -Wno-unused-result -I ../../../anyedit_amd/csrc ../pp_lab.hip"
for spec in "$@"; do
  tag=${spec%%=*}; flags=""; [ "$spec" != "$tag" ] && flags=${spec#*=}
  ( $B $flags -o pp_$tag > log_$tag.txt 2>&1 || echo "BUILD FAILED: $tag (see tools/ubench/build/log_$tag.txt)" ) &
done
wait
grep -l " error" log_*.txt 2>/dev/null
exit 0
