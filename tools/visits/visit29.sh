#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/v29_tmp.json 2> $OUT/v29_tmp.err; python -c "
import json; d=json.load(open('$OUT/v29_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')" || tail -5 $OUT/v29_tmp.err; }
run AE_PREFETCH=0
run AE_PREFETCH=1 AE_PREFETCH_DRY=1
run AE_PREFETCH=1 AE_PREFETCH_GROUP_MB=128
run AE_PREFETCH=1 AE_PREFETCH_GROUP_MB=128 AE_PREFETCH_DRY=1
run AE_PREFETCH=1 AE_PREFETCH_GROUP_MB=512 AE_PREFETCH_DRY=1
run AE_PREFETCH=0
