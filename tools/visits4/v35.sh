#!/bin/bash
# round 4 visit 35: the 128x128 LayerNorm-fold kernel pinned to 128 registers (two blocks per CU) against the unconstrained build (129 registers, variant
# -DAE_XE_OCC=0): A/B, then the lean evidence at the head — full GPU suite (recorded edit control), bench, rocprofv3 stats, traffic.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
T0=$(date +%s)
bash tools/ab_lib.sh noocc 3 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"value\": ([0-9.]+).*\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'img/s', round(float(m.group(2)) / 50, 3), 'ms per UNet step')
" | tee $OUT/r04_v35_xe_occ_ab.txt
echo "A/B done ($(( $(date +%s) - T0 )) s)"
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=5 ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - T0 )) s)"
grep -E "passed|failed" $OUT/pytest_gpu_full.log | tail -2
( timeout 600 python bench.py --steps 10 --warmup 2 ) > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cut -c1-330 $OUT/bench_full.json
( timeout 300 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"; cut -c1-200 $OUT/bench_default.json
cp $OUT/kernels_by_shape.json $OUT/kernels_by_shape_final.json 2>/dev/null
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/prof
bash tools/traffic.sh > $OUT/traffic.log 2>&1; echo "traffic rc=$?"
echo "total $(( $(date +%s) - T0 )) s"
