#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for km in 0 1 2; do
  for c in FETCH_SIZE; do
    ( cd /tmp && rm -rf /tmp/kt_$km && AE_CONV_KMAJOR=$km timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/kt_$km -o p -- python $R/tools/kbench.py "conv3x3 res" ) > $OUT/v22_pmc_$km.log 2>&1
    f=$(find /tmp/kt_$km -name "*counter_collection.csv" | head -1)
    python - "$f" $km <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# group by (kernel, grid) in dispatch order: kbench runs each case several times in a row
agg = collections.OrderedDict()
for r in rows:
    if "gemm_kernel" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"].split("(")[0][-60:], r.get("Grid_Size", r.get("Grid_Size_X","")), r.get("LDS_Block_Size",""))
    agg.setdefault(key, []).append(float(r["Counter_Value"]))
print("AE_CONV_KMAJOR=%s" % sys.argv[2])
for k, v in agg.items():
    print("   %-64s grid %-8s n=%3d FETCH_SIZE x2 = %8.1f MB per launch" % (k[0], k[1], len(v), 2.0 * 1024 * sum(v) / len(v) / 1e6))
PY
  done
  AE_CONV_KMAJOR=$km python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1|L3"
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v22_conv_fetch_by_shape.txt
