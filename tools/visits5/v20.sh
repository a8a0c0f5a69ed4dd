#!/bin/bash
# Round 5, visit 20: PMC of the self-attention backward passes (N = 4096, d = 40, B H = 32) alone: where do the wave cycles go?
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
python tools/attn_bwd_lab.py | tee $OUT/v20_attn_bwd.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  cd /tmp && rm -rf pmc_v20_$i && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_v20_$i -o p -- python $R/tools/attn_bwd_lab.py 32 4096 40 4 > $OUT/v20_pmc_$i.log 2>&1; echo "group $i rc=$?"
  cd $R
  python - /tmp/pmc_v20_$i <<'PY' | tee -a $OUT/v20_attn_bwd.txt
import csv, sys, glob, collections, re
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not cc:
    print("no counters in", d); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
    if "attn" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in agg.items():
    print(k)
    for c, v in sorted(dd.items()):
        print(f"   {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
PY
done
