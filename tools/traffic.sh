#!/bin/bash
# HBM traffic per kernel launch from the L2 memory-side counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE are
# collected in SEPARATE rocprofv3 --pmc passes (they do not fit one pass) over one un-graphed UNet evaluation loop of bench.py;
# no tracing domain other than --kernel-trace is enabled.  Output: gpurun_out/traffic.json (copy to profiles/).
# Corrections applied (and recorded in the file): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies the 128-B
# requests of wide (16 B/lane) coalesced reads at 64 B, so it is doubled.  layernorm_kernel (reads and writes exactly M*C*2 bytes)
# is the in-run calibration row: its corrected/expected ratio is printed next to every figure.
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/traffic_$c && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/traffic_$c -o t -- \
      python $R/bench.py --no-graph --ddim-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ) > $R/gpurun_out/traffic_$c.log 2>&1
  echo "$c rc=$?"
done
python - "$R" <<'PY'
import collections, csv, glob, json, re, sys
R = sys.argv[1]

def label(name):
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d)(?:, \d+, \d+, \w+, \d+, (\d+))?", name)
    if m:  # same family names as bench.py's folded profiler labels; the three-stage-ring instantiations are their own symbols
        ring = ",ring3" if m.group(4) == "3" else ""
        return f"gemm_kernel<{m.group(1)}x{m.group(2)}{ring},{'conv3x3' if m.group(3) == '1' else 'dense'}>"
    m = re.search(r"attn_(?:fast|pipe)_kernel<(\d+)", name)   # (round 5: the pipelined long-sequence kernel keeps the family label of bench.py's profiler)
    if m:
        return f"attn_fast_kernel<D={m.group(1)}>"
    m = re.search(r"gemm_rowpanel_kernel<(\d+), (\d+), (\w+)>", name)
    if m:  # <KS, EPI, LN>: K = 32 KS
        return f"gemm_rowpanel_kernel<K={32 * int(m.group(1))}{',LN' if m.group(3) in ('true', '1') else ''}{',geglu' if m.group(2) == '2' else ''}>"
    m = re.search(r"attn_kernel<(\d+)", name)
    if m:
        return f"attn_kernel<D={m.group(1)}>"
    m = re.search(r"(\w+_kernel|\w+)(<|\()", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1) if m else name[:60]

out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/traffic_{c}/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter file for", c)
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == c:
            agg[label(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = {"launches": len(v), "avg_raw": sum(v) / len(v)}
import subprocess
try:
    commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
if not commit and __import__("os").path.exists(f"{R}/.commit_id"):
    commit = open(f"{R}/.commit_id").read().strip()
res = {"_doc": "per-launch HBM-side bytes; raw counters are KiB; FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B)", "commit": commit, "kernels": {}}
for k, d in out.items():
    f = d.get("FETCH_SIZE", {}).get("avg_raw")
    w = d.get("WRITE_SIZE", {}).get("avg_raw")
    res["kernels"][k] = {"launches": d.get("FETCH_SIZE", d.get("WRITE_SIZE"))["launches"],
                         "fetch_bytes": None if f is None else 2.0 * 1024.0 * f, "write_bytes": None if w is None else 1024.0 * w,
                         "fetch_raw_kib": f, "write_raw_kib": w}
ln_name = "layernorm_rows_kernel" if "layernorm_rows_kernel" in res["kernels"] else "layernorm_kernel"   # round 3: the UNet's LayerNorms run the rows kernel
ln = res["kernels"].get(ln_name)
if ln:
    # per evaluation: 15 LN at [12288,640] + 15 at [3072,1280] (+ 3 at [768,1280]; the level-1 LayerNorms are fused into the row-panel
    # GEMM since round 2): expected mean
    exp = (15 * 12288 * 640 + 15 * 3072 * 1280 + 3 * 768 * 1280) * 2.0 / 33.0
    # round 4: with the LayerNorm fold (AE_LN_FOLD, default on) only the three [768, 1280] LayerNorms of the 8x8 level are still launches — told apart
    # by the launch count per evaluation (the d = 40 self-attention kernel runs five times per evaluation)
    att = res["kernels"].get("attn_fast_kernel<D=40>")
    if att and att["launches"] and ln["launches"] * 5.0 / att["launches"] < 8:
        exp = 768 * 1280 * 2.0
    res["calibration"] = {"kernel": ln_name, "expected_bytes_each_way_approx": exp,
                          "fetch_over_expected": ln["fetch_bytes"] / exp if ln["fetch_bytes"] else None,
                          "write_over_expected": ln["write_bytes"] / exp if ln["write_bytes"] else None}
json.dump(res, open(f"{R}/gpurun_out/traffic.json", "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["fetch_bytes"] or 0) * kv[1]["launches"])[:16]:
    print(f"{k:40s} n={v['launches']:5d} fetch={(v['fetch_bytes'] or 0) / 1e6:9.2f} MB write={(v['write_bytes'] or 0) / 1e6:9.2f} MB")
print(res.get("calibration"))
PY
