#!/bin/bash
# round 4 visit 8: ping-pong with the activation offsets prepared one LOAD interval ahead
cd tools/ubench/build
{
echo "=== plain, round-3 loops"; AE_GEMM_PP=0 ./pp_plain
echo "=== plain, PP"; AE_GEMM_PP=15 ./pp_plain x
echo "=== trace, PP"; AE_GEMM_PP=15 ./pp_trace c | head -30
echo "=== plain, PP again"; AE_GEMM_PP=15 ./pp_plain
} 2>&1 | tee ../../../gpurun_out/r04_v8_pp_prep.txt
