#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_abl.so
for a in 0 4 0 4; do
  echo "== AE_GEMM_ABL=$a (0 full, 4 = A tile DMA on 2 of 9 steps: slab-loader DMA volume)"
  AE_GEMM_ABL=$a python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1|L2" 
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v7b_gemm_ablation_slab.txt
