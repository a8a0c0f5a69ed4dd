#!/bin/bash
# Round 5, visit 23: ablations of the attention backward passes at d = 40 (lab library bwdlab, AE_BWD_ABL bits: 1 no second product, 2 no P block,
# 4 no first product, 8 no barrier / restage, 16 no global loads) — what bounds the pass?  rocprofv3 per-kernel times of each.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
: > $OUT/v23_ablations.txt
for abl in 0 1 2 4 8 16 24 3 6 7; do
  cd /tmp && rm -rf v23_$abl && AE_BWD_ABL=$abl AE_LIB_PATH=$R/anyedit_amd/libanyedit_hip_bwdlab.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/v23_$abl -o p -- python $R/tools/attn_bwd_lab.py 32 4096 40 10 > /tmp/v23_$abl.log 2>&1
  cd $R
  F=$(find /tmp/v23_$abl -name '*kernel_stats.csv' | head -1)
  python - "$F" $abl <<'PY' | tee -a $OUT/v23_ablations.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'attn_bwd_kernel' in r['Name']]
print('AE_BWD_ABL=%-3s' % sys.argv[2], '  '.join('%s %.1f us' % (r['Name'].split('attn_bwd_kernel')[1][:16], float(r['AverageNs']) / 1e3) for r in sorted(rows, key=lambda r: r['Name'])))
PY
done
