#!/bin/bash
# round 4 visit 17: ping-pong on a 256x128 tile for the 32x32-level launches (lab)
cd tools/ubench/build
{
echo "=== default (128x128 tile, two blocks per CU)"; ./pp_plain m
echo "=== 256x128 ping-pong, 4x2 waves"; AE_GEMM_PP=111 ./pp_plain m
echo "=== 256x128 ping-pong, 2x4 waves"; AE_GEMM_PP=111 ./pp_wm2 m
echo "=== default again"; ./pp_plain m
} 2>&1 | tee ../../../gpurun_out/r04_v17_pp256.txt
