#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or fuzz" ) > $OUT/v38_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v38_pytest.log | tail -3
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/sweep_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/sweep_tmp.json')); print('%-60s %.3f img/s  %.3f ms' % ('$*'[-60:], d['value'], d['unet_step_ms_p50']))"; }
run AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_prev.so
run AE_NEW=1
run AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_prev.so
run AE_NEW=1
