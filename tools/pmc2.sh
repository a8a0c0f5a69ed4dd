#!/bin/bash
# usage: tools/pmc2.sh <tag> "<kbench filter>" COUNTER...  -> gpurun_out/pmc_<tag>.txt
# One rocprofv3 --pmc pass (kernel-trace only, no other tracing domain) over tools/kbench.py cases; per kernel symbol: launches,
# average duration (from the dispatch timestamps of the same pass) and the average of every counter, plus — when GRBM_GUI_ACTIVE is
# among them — the effective shader clock GRBM_GUI_ACTIVE / duration (MI355X_MICROARCH.md "DVFS give-back").
tag=$1; flt=$2; shift 2
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
# filter "@bench": the counters over the un-graphed UNet evaluations of bench.py (every kernel of the step in its place in the sequence) instead of kbench cases
if [ "$flt" = "@bench" ]; then CMD="python $R/bench.py --no-graph --ddim-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline"; else CMD="python $R/tools/kbench.py \"$flt\""; fi
cd /tmp && rm -rf pmc_$tag && eval timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- $CMD > $R/gpurun_out/pmc_$tag.log 2>&1
echo "rocprofv3 rc=$?"
python - /tmp/pmc_$tag "$R/gpurun_out/pmc_$tag.txt" <<'PY'
import csv, sys, collections, glob, re
d, out = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc:
    print("no counter_collection.csv under", d); sys.exit(1)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        try:
            dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
        except Exception:
            pass
agg = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(cc[0])):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:96]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    did = r.get("Dispatch_Id")
    if did in dur and (k, did) not in seen:
        seen.add((k, did)); durs[k].append(dur[did][0])
    elif "Start_Timestamp" in r and (k, did) not in seen:
        try:
            seen.add((k, did)); durs[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        except Exception:
            pass
with open(out, "w") as f:
    for k, dd in agg.items():
        if not any(s in k for s in ("attn", "gemm", "gn_", "layernorm", "splitk", "colstats")):
            continue
        n = max(len(v) for v in dd.values())
        ns = sum(durs[k]) / len(durs[k]) if durs[k] else float("nan")
        f.write(f"{k}\n   launches={n}  avg_duration_us={ns / 1e3:.2f}\n")
        for c, v in sorted(dd.items()):
            f.write(f"   {c:32s} avg={sum(v) / len(v):16.1f}\n")
        if "GRBM_GUI_ACTIVE" in dd and durs[k]:
            g = sum(dd["GRBM_GUI_ACTIVE"]) / len(dd["GRBM_GUI_ACTIVE"])
            f.write(f"   effective_clock_GHz = GRBM_GUI_ACTIVE / duration = {g / ns:.3f}\n")
print(open(out).read())
PY
