#!/bin/bash
# round 4 visit 4: ping-pong lab — cycle buckets, ablations, DMA placement / priority variants (standalone binaries, no Python)
set -u
cd tools/ubench/build
OUT=../../../gpurun_out; mkdir -p $OUT
{
echo "=== plain build, PP=0 (round-3 loops)"; AE_GEMM_PP=0 ./pp_plain x
echo "=== plain build, PP=15"; AE_GEMM_PP=15 ./pp_plain x
echo "=== lab build (s_memtime stamps), PP=15"; AE_GEMM_PP=15 ./pp_lab
echo "=== ablation 1: no DMA after the prologue"; AE_GEMM_PP=15 ./pp_abl1
echo "=== ablation 2: DMA + barriers only"; AE_GEMM_PP=15 ./pp_abl2
echo "=== ablation 3: MFMAs on stale registers (no LDS reads)"; AE_GEMM_PP=15 ./pp_abl3
echo "=== DMA pieces in front of the fragment reads"; AE_GEMM_PP=15 ./pp_dmafirst
echo "=== no s_setprio"; AE_GEMM_PP=15 ./pp_noprio
echo "=== plain build again, PP=15"; AE_GEMM_PP=15 ./pp_plain
} 2>&1 | tee $OUT/r04_v4_pp_lab.txt
