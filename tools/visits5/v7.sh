#!/bin/bash
# Round 5, visit 7: which self-attention kernel is the default — AE_ATTN_V 3 (two-group) vs 7 (pipelined), in the graph (4 alternating pairs, 8 edits each) and isolated.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2 3 4; do
  for v in 3 7; do
    AE_ATTN_V=$v timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AE_ATTN_V=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v7_attn_default_ab.txt
for v in 3 7 3 7; do AE_ATTN_V=$v python tools/kbench.py "attn self N=4096" 2>/dev/null | grep "attn self" | sed "s/^/AE_ATTN_V=$v /"; done | tee -a $OUT/v7_attn_default_ab.txt
