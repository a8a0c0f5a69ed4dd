#!/bin/bash
# In-situ A/B of the tuning knobs against the defaults inside ONE visit: bench.py UNet step p50 per setting (3 timed edits each).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/sweep_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/sweep_tmp.json')); print('%-34s %.3f img/s  %.3f ms' % ('$*', d['value'], d['unet_step_ms_p50']))"; }
{
run AE_DEFAULT=1
run AE_CONV_T320_SPLITK=1
run AE_CONV_T320_SPLITK=3
run AE_GEMM_WK=1
run AE_GEMM_DEEP=0
run AE_CONV_DEEP=0
run AE_DEFAULT=1
run AE_GEMM_T320=15
run AE_GEMM_T320=3
run AE_GEMM_T160=0
run AE_GEMM_MFAST=0
run AE_LN_ROWS=0
run AE_DEFAULT=1
run AE_GN_COLSTATS=0
run AE_GEMM_DEEP64_MAX=256
run AE_ATTN_V=1
run AE_GEMM_W8=1
run AE_DEFAULT=1
} 2>&1 | tee $OUT/knob_sweep.txt
