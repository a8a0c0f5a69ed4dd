#!/bin/bash
# round 4 visit 5: M-split ping-pong (WA 4, AE_GEMM_PPM) vs K-split (WA 3, AE_GEMM_PP) vs round-3 loops
set -u
cd tools/ubench/build
OUT=../../../gpurun_out; mkdir -p $OUT
{
echo "=== plain, round-3 loops"; AE_GEMM_PP=0 ./pp_plain
echo "=== plain, K-split PP"; AE_GEMM_PP=15 ./pp_plain
echo "=== plain, M-split PPM"; AE_GEMM_PPM=7 ./pp_plain
echo "=== lab, PPM"; AE_GEMM_PPM=7 ./pp_lab
echo "=== PPM ablation 1: no DMA after the prologue"; AE_GEMM_PPM=7 ./pp_abl1
echo "=== PPM ablation 2: DMA + barriers only"; AE_GEMM_PPM=7 ./pp_abl2
echo "=== PPM ablation 3: no LDS reads"; AE_GEMM_PPM=7 ./pp_abl3
echo "=== plain, PPM again"; AE_GEMM_PPM=7 ./pp_plain
} 2>&1 | tee $OUT/r04_v5_ppm_lab.txt
