#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_AA=11 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py -m gpu -q -x -p no:cacheprovider -k "gemm or geglu or transformer or fuzz" ) > $OUT/v39_pytest.log 2>&1; echo "pytest AA=11 rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v39_pytest.log | tail -3
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/sweep_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/sweep_tmp.json')); print('%-24s %.3f img/s  %.3f ms' % ('$*', d['value'], d['unet_step_ms_p50']))
for k,v in json.load(open('$OUT/kernels_by_shape.json')).items():
    if '192x320,dense' in k and ('N=5120' in k or 'N=10240' in k): print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_GEMM_AA=3
run AE_GEMM_AA=11
run AE_GEMM_AA=3
run AE_GEMM_AA=11
