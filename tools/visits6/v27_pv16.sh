#!/bin/bash
# Self-attention d = 40: 48-row 16x16x32 PV products (AE_ATTN_PV16=0/1): isolated check + three alternating pairs of the bench
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out/v27; mkdir -p $OUT
timeout 600 python tools/attn_pipe_check.py 7 7p 2>&1 | grep -v -i "rccl\|amdgpu.ids" | tail -4 | tee $OUT/check.txt
for i in 1 2 3 4 5 6; do
  for f in 0 1; do
    echo "== AE_ATTN_PV16=$f (round $i)"
    AE_ATTN_PV16=$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  done
done | tee $OUT/ab.txt
