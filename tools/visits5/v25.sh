#!/bin/bash
# Round 5, visit 25: single-launch GroupNorm backward on the small maps (gnb_slab_kernel): tests, training A/B (AE_GN_BWD_SLAB).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_sam_anysd.py -m gpu -q -x -p no:cacheprovider -k "groupnorm or fuzz or train or grad" ) > $OUT/v25_pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/v25_pytest.log
for i in 1 2 3; do for v in 0 1; do echo -n "AE_GN_BWD_SLAB=$v: "; AE_GN_BWD_SLAB=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"; done; done 2>&1 | tee $OUT/v25_gnb_slab.txt
