#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_abl.so
for a in 0 10 11 12 13 14; do
  echo "== AE_GEMM_ABL=$a (0 shipped 2 blocks/CU; one block per CU: 10 two-stage full, 11 ring3 full, 12 two-stage DMA-only, 13 ring3 DMA-only, 14 two-stage no-DMA)"
  AE_GEMM_ABL=$a python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L2" 
  AE_GEMM_ABL=$a python tools/kbench.py "gemm " 2>&1 | grep -E "ff2 L2|qkv L2"
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v7c_ring_depth_ablation.txt
