#!/bin/bash
# Round 5, visit 5: narrow-output conv kernel (UNet head / VAE conv_out), time-embedding chain hoisted out of the DDIM loop: tests, bench A/Bs.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "== operator tests"
( timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "narrow or conv3x3" ) > $OUT/v5_pytest_ops.log 2>&1; echo "rc=$?"; tail -3 $OUT/v5_pytest_ops.log
echo "== model / pipeline tests"
( timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_sam_anysd.py tests/test_hip_fullsize.py -m gpu -q -x -s -p no:cacheprovider -k "not masked_edit_5_steps" ) > $OUT/v5_pytest_models.log 2>&1; echo "rc=$?"
grep -E "passed|failed|Error|kl-f8|edit" $OUT/v5_pytest_models.log | tail -10
echo "== bench A/B (alternating): AE_CONV_NARROW x AE_HOIST_TEMB"
for i in 1 2; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    AE_CONV_NARROW=$1 AE_HOIST_TEMB=$2 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NARROW=$1 HOIST=$2', d['value'], d['ms_per_step'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v5_bench_ab.txt
