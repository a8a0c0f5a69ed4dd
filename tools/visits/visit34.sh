#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/sweep_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/sweep_tmp.json')); print('%-34s %.3f img/s  %.3f ms' % ('$*', d['value'], d['unet_step_ms_p50']))"; }
{
run AE_DEFAULT=1
run AE_CONV_T320_SPLITK=0
run AE_CONV_T320_SPLITK=1
run AE_GEMM_DEEP=0
run AE_CONV_DEEP=0
run AE_DEFAULT=1
run AE_GEMM_T320=3
run AE_GEMM_T320=9
run AE_GEMM_T320=1
run AE_GEMM_T320=10
run AE_GEMM_AA=7
run AE_DEFAULT=1
} 2>&1 | tee $OUT/knob_sweep2.txt
