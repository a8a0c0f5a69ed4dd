#!/bin/bash
# round 4 visit 36: the 8-channel stem conv as im2col + K = 128 dense GEMM (AE_STEM_IM2COL, default on) against the implicit-GEMM form: operator test,
# alternating A/B, then the lean evidence at the head (full GPU suite, bench, rocprofv3 stats).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 300 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider -k "stem_conv or conv3x3" ) > $OUT/v36_ops.log 2>&1; echo "ops rc=$? ($(( $(date +%s) - T0 )) s)"; tail -1 $OUT/v36_ops.log
run() { echo -n "$1: "; env $1 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"; }
{ for r in 1 2 3; do run "AE_STEM_IM2COL=0"; run "AE_STEM_IM2COL=1"; done; } | tee $OUT/r04_v36_stem_im2col_ab.txt
echo "A/B done ($(( $(date +%s) - T0 )) s)"
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=5 ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - T0 )) s)"
grep -E "passed|failed" $OUT/pytest_gpu_full.log | tail -2
( timeout 600 python bench.py --steps 10 --warmup 2 ) > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cut -c1-330 $OUT/bench_full.json
cp $OUT/kernels_by_shape.json $OUT/kernels_by_shape_final.json 2>/dev/null
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/prof
echo "total $(( $(date +%s) - T0 )) s"
