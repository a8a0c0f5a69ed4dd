#!/usr/bin/env python3
"""Inter-kernel gaps of the captured UNet evaluation, from a rocprofv3 --kernel-trace CSV of bench.py (dev tool).
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [out.json]
Takes the trace's dispatches in start order, keeps the steady-state tail (the graph replays of the timed edit), and reports per UNet
step: sum of kernel durations, sum of the idle gaps between consecutive kernels, overlap, launches — i.e. what kernel boundaries
cost inside the HIP graph (MI355X_MICROARCH.md price list, row "boundary")."""
import csv
import glob
import json
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(files[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# one UNet evaluation starts with the timestep embedding kernel
starts = [i for i, r in enumerate(rows) if "timestep_embedding_kernel" in r[2]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    if len(seg) < 200:
        continue
    dur = sum(e - s for s, e, _ in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    steps.append({"launches": len(seg), "wall_us": (seg[-1][1] - seg[0][0]) / 1e3, "kernel_sum_us": dur / 1e3,
                  "idle_gap_sum_us": sum(g for g in gaps if g > 0) / 1e3, "overlap_sum_us": -sum(g for g in gaps if g < 0) / 1e3,
                  "gap_median_us": sorted(gaps)[len(gaps) // 2] / 1e3, "gap_p90_us": sorted(gaps)[int(len(gaps) * 0.9)] / 1e3})
tail = steps[-8:] if len(steps) >= 8 else steps
out = {"what": "per UNet evaluation inside the HIP-graph replay (rocprofv3 --kernel-trace; profiling inflates durations by a few percent)",
       "steps_seen": len(steps), "mean_of_last": {k: sum(s[k] for s in tail) / len(tail) for k in tail[0]} if tail else None, "last_steps": tail}
print(json.dumps(out["mean_of_last"], indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
