#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 170 python -m pytest tests/test_hip_unet.py tests/test_hip_sam_anysd.py -m gpu -q -p no:cacheprovider --durations=5 ) > $OUT/v44_pytest.txt 2>&1; echo "rc=$?"; tail -12 $OUT/v44_pytest.txt | cut -c1-200
