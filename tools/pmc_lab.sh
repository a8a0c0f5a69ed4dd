#!/bin/bash
# usage: tools/pmc_lab.sh <tag> "<env assignments>" <binary under tools/ubench/build> [args]   -> gpurun_out/pmclab_<tag>.txt
# Runs the standalone lab binary under rocprofv3 --pmc once per counter group below (kernel-trace only) and prints, per kernel symbol, the
# average of every counter over its dispatches.  The groups respect the per-block slot limits of gfx950 (MI355X_MICROARCH.md).
tag=$1; envs=$2; bin=$3; shift 3
export TMPDIR=/tmp
R=$PWD; mkdir -p $R/gpurun_out
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TD_SPI_STALL_sum TD_LOAD_WAVEFRONT_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
)
cd /tmp
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/pl_${tag}_$i
  env $envs timeout 120 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pl_${tag}_$i -o p -- $R/tools/ubench/build/$bin "$@" > /tmp/pl_${tag}_$i.log 2>&1 || echo "group $i ($g): rocprofv3 rc=$?"
  i=$((i+1))
done
python3 - $tag $i "$R/gpurun_out/pmclab_$tag.txt" <<'PY'
import csv, sys, collections, glob, re
tag, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for i in range(n):
    d = f"/tmp/pl_{tag}_{i}"
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc:
        print("no counters in group", i, open(f"/tmp/pl_{tag}_{i}.log").read()[-400:]); continue
    for r in csv.DictReader(open(cc[0])):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:80]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if kt:
        for r in csv.DictReader(open(kt[0])):
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:80]
            durs[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out, "w") as f:
    for k, dd in agg.items():
        ns = sum(durs[k]) / max(len(durs[k]), 1)
        f.write(f"{k}\n   avg_duration_us={ns / 1e3:.2f} (profiled passes)\n")
        for c, v in sorted(dd.items()):
            f.write(f"   {c:36s} avg={sum(v) / len(v):16.1f}  (n={len(v)})\n")
print(open(out).read())
PY
