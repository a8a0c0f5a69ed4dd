#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_WA=3 timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats or fuzz" ) > $OUT/v30_pytest.log 2>&1; echo "pytest WA=3 rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v30_pytest.log | tail -4
for wa in 0 3; do echo "== AE_GEMM_WA=$wa"; AE_GEMM_WA=$wa python tools/cold_weight_probe.py 2>&1 | grep -E "L2|L4|proj L3|launch"; done 2>&1 | grep -v amdgpu.ids | tee $OUT/v30_wa_probe.txt
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v30_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/v30_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')"; }
run AE_GEMM_WA=0
run AE_GEMM_WA=3
run AE_GEMM_WA=1
run AE_GEMM_WA=2
run AE_GEMM_WA=0
run AE_GEMM_WA=3
