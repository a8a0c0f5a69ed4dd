#!/bin/bash
# Round 5, visit 4: the ResBlock's skip 1x1 GEMM on a side stream (AE_SKIP_STREAM=1): UNet tests under it, bench A/B.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "== UNet tests with the side stream"
( AE_SKIP_STREAM=1 timeout 900 python -m pytest tests/test_hip_unet.py "tests/test_hip_bench_shapes.py::test_unet_bench_batch_vs_oracle_with_bf16_control" -m gpu -q -x -s -p no:cacheprovider ) > $OUT/v4_pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|UNet batch|Error" $OUT/v4_pytest.log | tail -6
echo "== bench A/B (alternating)"
for i in 1 2 3; do
  for v in 0 1; do
    AE_SKIP_STREAM=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AE_SKIP_STREAM=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v4_bench_ab.txt
