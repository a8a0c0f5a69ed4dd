#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "ms_deform" ) > $OUT/v42_msda.log 2>&1; echo "msda rc=$?"; tail -12 $OUT/v42_msda.log | cut -c1-220
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/v42_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/v42_smoke.log | cut -c1-200
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $OUT/v42_pytest_gpu.txt 2>&1; echo "full rc=$?"; tail -4 $OUT/v42_pytest_gpu.txt | cut -c1-200
git rev-parse HEAD 2>/dev/null
