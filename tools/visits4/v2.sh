#!/bin/bash
# round 4 visit 2: A/B of the AE_CONV_SPEC build (round-3 leftover), full JSON lines kept
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
C=$R/anyedit_amd/libanyedit_hip_cspec.so
for i in 1 2 3; do
  for v in product cspec; do
    if [ $v = product ]; then L=""; else L=$C; fi
    echo -n "$v round $i: "
    AE_LIB_PATH=$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"
  done
done | tee $OUT/r04_v2_cspec_ab.txt
