#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_AA=7 timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats or fuzz" ) > $OUT/v33_pytest.log 2>&1; echo "pytest AA=7 rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v33_pytest.log | tail -4
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v33_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v33_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')
for k,v in json.load(open('$OUT/kernels_by_shape.json')).items():
    if ('splitK,conv3x3' in k and ('M=3072 Cin=1280 Cout=1280 s1' in k or 'M=768 Cin=1280 Cout=1280 s1' in k)) or ('ring3,dense' in k and 'K=1280' in k and 'M=3072' in k): print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_GEMM_AA=3
run AE_GEMM_AA=7
run AE_GEMM_AA=3
run AE_GEMM_AA=7
