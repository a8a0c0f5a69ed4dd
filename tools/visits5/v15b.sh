#!/bin/bash
# Round 5, visit 15b: rocprofv3 kernel totals of the training step (csv output).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v15_prof -o train -- python $R/tools/bench_train.py --steps 8 --warmup 2 ) > $OUT/v15_rocprof.log 2>&1; echo "rc=$?"
cd $R
F=$(find $OUT/v15_prof -name '*kernel_stats.csv' | head -1)
cp "$F" $OUT/v15_train_kernel_stats.csv 2>/dev/null
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/v15_train_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('all kernels: %.1f ms over %d launches (10 steps + setup)' % (tot / 1e6, calls))
for r in rows[:70]:
    print('%8.3f ms %6s x %8.1f us  %s' % (int(r['TotalDurationNs']) / 1e6 / 10, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:170]))
PY
rm -rf $OUT/v15_prof
