#!/bin/bash
# Round 5, visit 11: re-test the split-K plan knob for the long-K 32x32-level convs under the ping-pong loop (AE_CONV_T320_SPLITK 2 = default, 3 = also >= 180 K tiles, 1 = every 128-tile grid).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2; do
  for v in 2 3 1; do
    AE_CONV_T320_SPLITK=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T320_SPLITK=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v11_t320_splitk_ab.txt
