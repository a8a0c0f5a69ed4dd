#!/bin/bash
# round 4 visit 6: PMC passes over the lab binary (conv 960->320 @64): round-3 loop vs K-split ping-pong
set -u
bash tools/pmc_lab.sh r3loop "AE_GEMM_PP=0" pp_plain c > gpurun_out/r04_v6_pmc_r3loop.log 2>&1
bash tools/pmc_lab.sh pp "AE_GEMM_PP=15" pp_plain c > gpurun_out/r04_v6_pmc_pp.log 2>&1
tail -5 gpurun_out/r04_v6_pmc_pp.log
