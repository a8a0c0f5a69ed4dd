#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_AA=3 timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats or fuzz" ) > $OUT/v32_pytest.log 2>&1; echo "pytest AA=3 rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v32_pytest.log | tail -4
for wa in 0 3; do echo "== AE_GEMM_WA=$wa"; AE_GEMM_WA=$wa python tools/cold_weight_probe.py 2>&1 | grep -E "conv res L2|conv res dec L2|launch"; done 2>&1 | grep -v amdgpu.ids
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v32_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v32_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')
for k,v in json.load(open('$OUT/kernels_by_shape.json')).items():
    if ('192x320' in k and 'M=49152' in k and 'conv' in k) or ('128x128,conv3x3' in k and 'Cin=640 Cout=640' in k): print('   ',k, v['calls'], round(v['avg_us'],1))
"; }
run AE_GEMM_WA=0 AE_GEMM_AA=0
run AE_GEMM_WA=3 AE_GEMM_AA=1
run AE_GEMM_WA=3 AE_GEMM_AA=3
run AE_GEMM_WA=0 AE_GEMM_AA=0
run AE_GEMM_WA=3 AE_GEMM_AA=1
run AE_GEMM_WA=3 AE_GEMM_AA=3
