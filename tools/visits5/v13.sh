#!/bin/bash
# Round 5, visit 13: LayerNorm fold on the row-panel kernel (64x64 level, AE_RP_FOLD): operator tests, transformer / UNet tests, bench A/B.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -s -p no:cacheprovider -k "layernorm_folded or rowpanel or ln_gemm or transformer" ) > $OUT/v13_pytest_ops.log 2>&1; echo "rc=$?"
grep -E "passed|failed|LN fold M=49|Error" $OUT/v13_pytest_ops.log | tail -8
( timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_bench_shapes.py -m gpu -q -x -s -p no:cacheprovider -k "unet" ) > $OUT/v13_pytest_unet.log 2>&1; echo "rc=$?"
grep -E "passed|failed|UNet batch|Error" $OUT/v13_pytest_unet.log | tail -5
run() { env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"; }
for i in 1 2 3; do run AE_RP_FOLD=0; run AE_RP_FOLD=1; done | tee $OUT/v13_ab.txt
for v in 0 1; do
  AE_RP_FOLD=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - $v <<'PY'
import json, sys
d=json.load(open('gpurun_out/kernels_by_shape.json'))
for k,v in d.items():
    if 'rowpanel' in k: print('RP_FOLD=' + sys.argv[1], k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
done | tee $OUT/v13_rowpanel_table.txt
