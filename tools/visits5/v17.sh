#!/bin/bash
# Round 5, visit 17: query-split dK / dV pass for few-key attention backward + 16-byte rowsum: tests, training step, add census.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider ) > $OUT/v17_pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/v17_pytest.log
for i in 1 2; do
  for v in 0 1; do
    echo "== AE_ATTN_BWD_SPLIT=$v"; AE_ATTN_BWD_SPLIT=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"
  done
done 2>&1 | tee $OUT/v17_train_ab.txt
( timeout 300 python tools/train_add_census.py ) > $OUT/v17_add_census.txt 2>&1; echo "rc=$?"; tail -60 $OUT/v17_add_census.txt
