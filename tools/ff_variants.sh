#!/bin/bash
# Lab builds of the fused feed-forward kernel: relinks the product library's objects with ff_fused.hip recompiled under extra -D flags.
#   tools/ff_variants.sh tag "-DFF_LAB=1" [tag2 "-D..."] ...   ->  anyedit_amd/libanyedit_hip_<tag>.so   (select with AE_LIB_PATH)
set -e
cd "$(dirname "$0")/../anyedit_amd"
python -m anyedit_amd.build >/dev/null 2>&1 || (cd .. && python -m anyedit_amd.build >/dev/null)
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_lab
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize $flags -c csrc/ff_fused.hip -o build_lab/ff_fused_$tag.o
  objs=$(ls build/*.o | grep -v ff_fused.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o libanyedit_hip_$tag.so $objs build_lab/ff_fused_$tag.o
  echo built libanyedit_hip_$tag.so
done
