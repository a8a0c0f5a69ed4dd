#!/bin/bash
# Round 5, visit 8: the LayerNorm-fold qkv of the 16x16 level on the 192x320 tile (AE_GEMM_T320_XE): parity tests, per-launch time in the eager table, bench A/B.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py -m gpu -q -x -s -p no:cacheprovider -k "layernorm_folded or transformer or unet" ) > $OUT/v8_pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|LN fold M=3072 C=1280 N=3840|Error" $OUT/v8_pytest.log | tail -6
for v in 0 1; do
  AE_GEMM_T320_XE=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - $v <<'PY'
import json, sys
d=json.load(open('gpurun_out/kernels_by_shape.json'))
for k,v in d.items():
    if 'N=3840' in k: print('T320_XE=' + sys.argv[1], k, v)
PY
done | tee $OUT/v8_qkv_l3_timing.txt
for i in 1 2 3; do
  for v in 0 1; do
    AE_GEMM_T320_XE=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T320_XE=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v8_bench_ab.txt
