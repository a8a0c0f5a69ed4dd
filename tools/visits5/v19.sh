#!/bin/bash
# Round 5, visit 19: 64x320 ping-pong conv tile for the training batch (M = 16384): parity green (1.66e-3), training step +0.17 ms in three alternating pairs
# (profiles/r05_v19_conv_t64x320_train_ab.txt) -> the instantiation, its plan rule (AE_CONV_T64) and its test were removed again; this script is the record.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_bench_shapes.py -m gpu -q -x -s -p no:cacheprovider -k "64x320 or training_step" ) > $OUT/v19_pytest.log 2>&1; echo "rc=$?"; grep -E "rel-L2|passed|failed|Error|assert" $OUT/v19_pytest.log | tail -20
for i in 1 2 3; do
  for v in 0 1; do
    echo "== AE_CONV_T64=$v"; AE_CONV_T64=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"
  done
done 2>&1 | tee $OUT/v19_train_ab.txt
