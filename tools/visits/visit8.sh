#!/bin/bash
# Round-3 visit 8: whole -m gpu suite at the current defaults (producer statistics on, attention V=3, multi-row LayerNorm), bench.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=12 ) > $OUT/v8_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" $OUT/v8_pytest.log | tail -12
( timeout 900 python bench.py --steps 6 --warmup 2 ) > $OUT/v8_bench.json 2> $OUT/v8_bench.err; echo "bench rc=$?"; cut -c1-900 $OUT/v8_bench.json
cp $OUT/kernels_by_shape.json $OUT/v8_kernels_by_shape.json
( AE_LN_ROWS=0 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v8_bench_ln0.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v8_bench_ln0.json')); print('AE_LN_ROWS=0:', round(d['value'],3), round(d['unet_step_ms_p50'],3))"
( timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v8_bench_b.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v8_bench_b.json')); print('default again:', round(d['value'],3), round(d['unet_step_ms_p50'],3))"
( timeout 600 python tools/bench_train.py --steps 6 --warmup 2 ) > $OUT/v8_train.json 2> $OUT/v8_train.err; cut -c1-400 $OUT/v8_train.json
