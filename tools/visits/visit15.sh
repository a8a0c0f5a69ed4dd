#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_sam_anysd.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "two_segments or anysd or moe or unet_bench or training" ) > $OUT/v15_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/v15_pytest.log
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v15_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v15_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')"; }
run AE_ATTN_SEG2_160=0
run AE_ATTN_SEG2_160=1
run AE_ATTN_SEG2_160=0
run AE_ATTN_SEG2_160=1
