#!/bin/bash
# usage: tools/pmc.sh <tag> "<kbench filter>" COUNTER...   -> gpurun_out/pmc_<tag>.csv (per-kernel averages)
tag=$1; flt=$2; shift 2
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf pmc_$tag && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/kbench.py "$flt" > $R/gpurun_out/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" "$R/gpurun_out/pmc_$tag.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, d in agg.items():
        if "attn" in k or "gemm" in k or "gn_" in k or "ff_fused" in k:
            f.write(k + "\n")
            for c, v in sorted(d.items()):
                f.write(f"   {c:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}\n")
print(open(sys.argv[2]).read())
PY
