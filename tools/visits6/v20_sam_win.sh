#!/bin/bash
# SAM windowed blocks: LayerNorm with the window addressing folded in (AE_SAM_WIN_FUSED=0/1), bit-exact test, encoder A/B, kernel trace of the encoder.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out/v20; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_sam_anysd.py -q -m gpu -x -k "window or sam or encoder" 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python -m pytest tests/test_hip_bench_shapes.py -q -m gpu -x -k "sam" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2 3; do
  for f in 0 1; do
    echo "== AE_SAM_WIN_FUSED=$f (round $i)"
    AE_SAM_WIN_FUSED=$f timeout 300 python tools/bench_sam.py --iters 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['latency_ms_p50'], d['latency_ms_p50_hip_graph'])"
  done
done | tee $OUT/ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o sam -- python $R/tools/bench_sam.py --iters 10 > /dev/null 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/sam_kernel_stats.csv; head -40 $OUT/sam_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
