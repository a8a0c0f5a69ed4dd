#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_hip_unet.py -m gpu -q -x -p no:cacheprovider -k "dpm" ) > $OUT/v40_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/v40_pytest.log | cut -c1-200
