#!/bin/bash
# Round 5, visit 15: where the training step's time goes: host enqueue against step time, by-shape table, rocprofv3 kernel totals.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 300 python tools/bench_train.py --steps 6 --warmup 2 --per-step ) > $OUT/v15_train_per_step.json 2> $OUT/v15_train_per_step.err; echo "rc=$?"
( AE_TRAIN_PROFILE=1 timeout 300 python tools/bench_train.py --steps 4 --warmup 2 ) > $OUT/v15_train_profile.json 2> $OUT/v15_train_profile.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/v15_train_per_step.json').read().strip().splitlines()[-1])
print('step', round(d['ms_per_step'], 2), 'ms; (host enqueue, step) per step:', d['per_step_ms'])
d = json.loads(open('gpurun_out/v15_train_profile.json').read().strip().splitlines()[-1])
p = d['profile']
print('families:', json.dumps(p['families']))
for k, v in list(p['shapes'].items())[:40]:
    print(f"  {v['ms']:7.3f} ms  {v['calls']:4d} x {v['avg_us']:7.1f} us  {v['tflops']:6.1f} TF  {k}")
PY
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/v15_prof -o train -- python $R/tools/bench_train.py --steps 6 --warmup 2 ) > $OUT/v15_rocprof.log 2>&1; echo "rc=$?"
cd $R
F=$(find $OUT/v15_prof -name '*kernel_stats.csv' | head -1)
cp "$F" $OUT/v15_train_kernel_stats.csv 2>/dev/null
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/v15_train_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('all kernels: %.1f ms over %d launches (8 steps + setup)' % (tot / 1e6, calls))
for r in rows[:45]:
    print('%8.3f ms %6s x %8.1f us  %s' % (int(r['TotalDurationNs']) / 1e6 / 8, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:150]))
PY
rm -rf $OUT/v15_prof
