#!/bin/bash
# SAM window attention: compact LDS map (70 KB per block) — tests, then the encoder with one block per CU (187 registers) and two (128-register cap, spills)
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out/v29; mkdir -p $OUT
for o in 0 1; do
  AE_ATTN_WIN_OCC4=$o timeout 600 python -m pytest tests/test_hip_sam_anysd.py tests/test_hip_bench_shapes.py -q -m gpu -x -k "relpos or sam" 2>&1 | grep -E "passed|failed|error" | tail -2
done
for i in 1 2 3; do
  for f in 0 1; do
    echo "== AE_ATTN_WIN_OCC4=$f (round $i)"
    AE_ATTN_WIN_OCC4=$f timeout 300 python tools/bench_sam.py --iters 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['latency_ms_p50'], d['latency_ms_p50_hip_graph'], {k:round(v['avg_us'],1) for k,v in d['by_shape'].items() if 'Nq=196' in k})"
  done
done | tee $OUT/ab.txt
