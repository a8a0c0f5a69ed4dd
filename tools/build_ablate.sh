#!/bin/bash
# Lab build of the library with the main-loop ablations of gemm_conv.hip compiled in (AE_GEMM_ABL=1|2|3 at run time):
#   anyedit_amd/build_abl/libanyedit_hip_abl.so; use it with AE_LIB_PATH.  Results of ablated kernels are WRONG by construction.
set -e
cd "$(dirname "$0")/.."
O=anyedit_amd/build_abl; mkdir -p $O
for f in c_api gemm_rowpanel attention attention_fast attention_fp8 attention_bwd norm elementwise backward gate expert_kv msda sam_decoder; do
  cp anyedit_amd/build/$f.o $O/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DAE_GEMM_ABLATE $AE_ABL_EXTRA -c anyedit_amd/csrc/gemm_conv.hip -o $O/gemm_conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libanyedit_hip_abl.so $O/*.o
echo built $O/libanyedit_hip_abl.so
