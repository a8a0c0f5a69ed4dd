#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v23_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v23_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')
for k,v in json.load(open('$OUT/kernels_by_shape.json')).items():
    if 'conv3x3' in k and ('M=12288' in k) : print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_CONV_KMAJOR=1
run AE_CONV_KMAJOR=4
run AE_CONV_KMAJOR=1
run AE_CONV_KMAJOR=4
