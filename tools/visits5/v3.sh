#!/bin/bash
# Round 5, visit 3: Upsample as four 2x2 convs (ae_conv3x3_up2_bf16) — operator tests, the UNet / VAE tests that now run through it, bench A/B by knob.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "== operator tests"
( timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "up2 or conv3x3 or producer_colstats or attention_pipelined" ) > $OUT/v3_pytest_ops.log 2>&1; echo "rc=$?"; tail -3 $OUT/v3_pytest_ops.log
echo "== model tests"
( timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py tests/test_hip_bench_shapes.py -m gpu -q -x -s -p no:cacheprovider -k "not masked_edit_5_steps and not training_step and not launcher" ) > $OUT/v3_pytest_models.log 2>&1; echo "rc=$?"
grep -E "passed|failed|rel-L2|HIP .* control|Error" $OUT/v3_pytest_models.log | tail -14
echo "== bench A/B (alternating)"
for i in 1 2; do
  for v in 0 1; do
    AE_UP2_SUBPIXEL=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AE_UP2_SUBPIXEL=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v3_bench_ab.txt
echo "== kernels by shape"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/v3_bench_line.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open('gpurun_out/kernels_by_shape.json'))
for k,v in d.items():
    if 'up2' in k or 's1u' in k: print(k, v)
PY
