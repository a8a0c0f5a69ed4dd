#!/bin/bash
# round 4 visit 7: barrier-arrival timeline of all eight waves of one block (K-split ping-pong, conv 960->320)
cd tools/ubench/build
AE_GEMM_PP=15 ./pp_trace c 2>&1 | tee ../../../gpurun_out/r04_v7_pp_trace.txt
