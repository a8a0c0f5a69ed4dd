#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/sweep_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/sweep_tmp.json')); print('%-40s %.3f img/s  %.3f ms' % ('$*', d['value'], d['unet_step_ms_p50']))"; }
{
run AE_DEFAULT=1
run AE_CONV_DEEP=0
run AE_GEMM_AA=7
run AE_CONV_DEEP=0 AE_GEMM_AA=7
run AE_DEFAULT=1
run AE_CONV_DEEP=0
run AE_GEMM_AA=7
run AE_CONV_DEEP=0 AE_GEMM_AA=7
} 2>&1 | tee -a $OUT/knob_sweep2.txt
