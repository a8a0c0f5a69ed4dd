#!/bin/bash
# Round 5, visit 26: attention backward with the MFMA phase fences (+ one query fragment for the two-accumulator dQ pass at d = 64 / 80): tests incl. 30-launch
# determinism, lab timings, training step.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_sam_anysd.py -m gpu -q -x -p no:cacheprovider -k "attention or fuzz or train or grad" ) > $OUT/v26_pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/v26_pytest.log
for shape in "32 4096 40" "32 1024 80" "32 256 160"; do python tools/attn_bwd_lab.py $shape 20; done 2>&1 | grep "attention backward" | tee $OUT/v26_attn_bwd.txt
for i in 1 2 3; do timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"; done 2>&1 | tee -a $OUT/v26_attn_bwd.txt
