#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v28_tmp.json 2> $OUT/v28_tmp.err; python -c "
import json; d=json.load(open('$OUT/v28_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms', d.get('parity'))" || tail -5 $OUT/v28_tmp.err; }
run AE_PREFETCH=0
run AE_PREFETCH=1
run AE_PREFETCH=0
run AE_PREFETCH=1
( timeout 600 python -m pytest tests/test_hip_sam_anysd.py tests/test_hip_unet.py -m gpu -q -x -p no:cacheprovider -k "pipeline or edit or sampler or graph" ) > $OUT/v28_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v28_pytest.log | tail -5
