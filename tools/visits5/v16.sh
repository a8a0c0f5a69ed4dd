#!/bin/bash
# Round 5, visit 16: kernel trace of the training step (per-dispatch rows: gaps between kernels, grid sizes) + who calls torch copies.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/v16_prof -o train -- python $R/tools/bench_train.py --steps 3 --warmup 2 ) > $OUT/v16_rocprof.log 2>&1; echo "rc=$?"
cd $R
F=$(find $OUT/v16_prof -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
# keep: name (short), start, end, grid, workgroup
out = open('gpurun_out/v16_train_trace.tsv', 'w')
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    out.write('%s\t%d\t%d\t%s\t%s\n' % (r['Kernel_Name'][:140], int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
out.close()
PY
gzip -f $OUT/v16_train_trace.tsv
rm -rf $OUT/v16_prof
( AE_TRAIN_COPYTRACE=1 timeout 300 python tools/train_copytrace.py ) > $OUT/v16_copytrace.txt 2>&1; echo "rc=$?"; tail -40 $OUT/v16_copytrace.txt
