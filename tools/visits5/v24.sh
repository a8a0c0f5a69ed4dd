#!/bin/bash
# Round 5, visit 24: split-K cap of the small-grid convs (AE_CONV_SPLIT_MAX 8 vs 16: the M = 256 convs of a training batch are 20 tiles of 128x128) in the training step.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2 3; do for v in 8 16; do echo -n "AE_CONV_SPLIT_MAX=$v: "; AE_CONV_SPLIT_MAX=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"; done; done 2>&1 | tee $OUT/v24_split_max.txt
