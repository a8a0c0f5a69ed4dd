#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 480 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/v43_pytest_gpu.txt 2>&1; echo "full rc=$?"; tail -6 $OUT/v43_pytest_gpu.txt | cut -c1-200
