#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in 0 1; do
cd /tmp && rm -rf v28_$v && AE_ATTN_BWD_PIPE=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v28_$v -o p -- python $R/tools/attn_bwd_lab.py 32 4096 40 10 > /tmp/v28_$v.log 2>&1
cd $R
F=$(find /tmp/v28_$v -name '*kernel_stats.csv' | head -1)
python - "$F" $v <<'PY' | tee -a $OUT/v28_bwd_pipe.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'attn_bwd_kernel' in r['Name']]
print('AE_ATTN_BWD_PIPE=%s' % sys.argv[2], '  '.join('%s %.1f us' % (r['Name'].split('attn_bwd_kernel')[1][:26], float(r['AverageNs']) / 1e3) for r in sorted(rows, key=lambda r: r['Name'])))
PY
done
