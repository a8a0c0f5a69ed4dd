#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py tests/test_hip_unet.py -m gpu -q -x -s -p no:cacheprovider -k "conv3x3 or unet_bench_batch or fuzz or resblock or tiny" ) > $OUT/v21_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert|chunk-major" $OUT/v21_pytest.log | tail -12
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v21_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v21_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')
for k,v in json.load(open('$OUT/kernels_by_shape.json')).items():
    if 'conv3x3' in k and ('M=49152' in k) : print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_CONV_KMAJOR=0
run AE_CONV_KMAJOR=1
run AE_CONV_KMAJOR=2
run AE_CONV_KMAJOR=3
run AE_CONV_KMAJOR=0
run AE_CONV_KMAJOR=1
bash tools/traffic.sh > $OUT/v21_traffic.log 2>&1; tail -12 $OUT/v21_traffic.log
