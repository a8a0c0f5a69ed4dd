#!/bin/bash
# Round 5, visit 21 (re-used for 21b: staging addresses hoisted + packed fp32 math in the P computation, against the commit before):
# (variant library bwdold = the commit before): tests, the lab shape, PMC conflict counter, training step.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider -k "attention or fuzz" ) > $OUT/v21_pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/v21_pytest.log
for shape in "32 4096 40" "32 1024 80" "32 256 160"; do
  bash tools/ab_lib.sh bwdold 2 python tools/attn_bwd_lab.py $shape 20
done 2>&1 | tee $OUT/v21_attn_bwd_ab.txt
bash tools/ab_lib.sh bwdold 3 python tools/bench_train.py --steps 8 --warmup 2 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'ms per training step')
" | tee -a $OUT/v21_attn_bwd_ab.txt
cd /tmp && rm -rf pmc_v21 && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_v21 -o p -- python $R/tools/attn_bwd_lab.py 32 4096 40 4 > $OUT/v21_pmc.log 2>&1; echo "pmc rc=$?"
cd $R
python - /tmp/pmc_v21 <<'PY' | tee -a $OUT/v21_attn_bwd_ab.txt
import csv, sys, glob, collections, re
cc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
    if "attn_bwd" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in agg.items():
    print(k)
    for c, v in sorted(dd.items()):
        print(f"   {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
PY
