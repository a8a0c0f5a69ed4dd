#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v10_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v10_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')"; }
run AE_ATTN_V=3
run AE_ATTN_V=1
run AE_ATTN_V=3 AE_GEMM_T320=11
run AE_ATTN_V=3
run AE_ATTN_V=1
run AE_ATTN_V=3 AE_GEMM_T320=11
run AE_ATTN_V=3 AE_GN_COLSTATS=0
