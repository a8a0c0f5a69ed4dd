#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_backward.py tests/test_hip_unet.py -m gpu -q -x -p no:cacheprovider -k "attention or fuzz or two_segments or transformer" ) > $OUT/v16_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $OUT/v16_pytest.log | tail -3
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v16_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v16_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')
for k,v in d.get('kernels',{}).items():
    if 'D=160' in k: print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_ATTN_FAST160=0
run AE_ATTN_FAST160=1
run AE_ATTN_FAST160=0
run AE_ATTN_FAST160=1
