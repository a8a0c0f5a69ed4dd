#!/bin/bash
# Round 5, visit 12: split-K 192x320 plan for the long-K 32x32-level convs as the default (slab form through the mirrored K order): tests, A/B against knob 2, a few other stale knobs.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_bench_shapes.py tests/test_hip_unet.py -m gpu -q -x -s -p no:cacheprovider -k "long_k or bench_plan or unet_bench_batch or unet" ) > $OUT/v12_pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|split-K 192x320|UNet batch|Error" $OUT/v12_pytest.log | tail -8
run() { env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"; }
for i in 1 2 3; do
  run AE_CONV_T320_SPLITK=2
  run AE_CONV_T320_SPLITK=3
done | tee $OUT/v12_ab.txt
for k in AE_CONV_T320_SPLITK=1 AE_CONV_DEEP=1 AE_GEMM_T320=15 AE_GEMM_WK=1 AE_GN_FUSE=1 AE_CONV_T320_SPLITK=3; do run $k; done | tee -a $OUT/v12_ab.txt
