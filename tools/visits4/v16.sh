#!/bin/bash
# round 4 visit 16: training step and SAM encoder with / without the ping-pong GEMM loop
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
{ for pp in 0 15 0 15; do echo -n "train AE_GEMM_PP=$pp: "; AE_GEMM_PP=$pp timeout 200 python tools/bench_train.py --steps 10 --warmup 2 2>/dev/null | tail -1 | cut -c1-260; done
  for pp in 0 15; do echo "sam AE_GEMM_PP=$pp: "; AE_GEMM_PP=$pp timeout 200 python tools/bench_sam.py 2>/dev/null | tail -3 | cut -c1-400; done
} | tee $OUT/r04_v16_train_sam_pp.txt
