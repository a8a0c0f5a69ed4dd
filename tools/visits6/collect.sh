#!/bin/bash
# Copies what tools/visits6/evidence.sh left in gpurun_out/ into profiles/r06_<tag>_* (run in the build container after the visit).  usage: bash tools/visits6/collect.sh final5
set -eu
tag=$1; O=gpurun_out; P=profiles/r06_${tag}
cp $O/bench_full.json ${P}_bench.json; cp $O/bench_default.json ${P}_bench_default.json
cp $O/ff_fused_lab.json ${P}_ff_fused_lab.json; cp $O/xattn_fused_lab.json ${P}_xattn_fused_lab.json
cp $O/gemm_vs_library.txt ${P}_gemm_vs_library.txt; cp $O/sam_gemm_vs_library.txt ${P}_sam_gemm_vs_library.txt
cp $O/kernel_stats.csv ${P}_kernel_stats.csv; cp $O/sam_kernel_stats.csv ${P}_sam_kernel_stats.csv; cp $O/kernels_by_shape_final.json ${P}_kernels_by_shape.json
for k in attn attnx conv dense ff gn; do   # the per-kernel counter averages (pmc_*_a.csv) first, then the run's own lines of the log
  { echo "# rocprofv3 --pmc pass over tools/kbench.py (tools/pmc.sh ${k}_a ...): per-kernel averages of the counters, then the run's log"; cat $O/pmc_${k}_a.csv; echo
    grep -v "simple_timer\|output_stream\|tool.cpp" $O/pmc_${k}_a.log || true; } > ${P}_pmc_${k}.txt
done
grep -v "DeprecationWarning\|^  \|^tests/.*::\|^$" $O/pytest_gpu_full.log | cut -c1-400 > ${P}_pytest_gpu.txt
cp $O/sam_encoder.json ${P}_sam_encoder.json; cp $O/smi_during_bench.json ${P}_smi_during_bench.json; cp $O/smoke.log ${P}_smoke.txt
cp $O/throttle_bench_final.json ${P}_throttle_bench.json; cp $O/train_sweep.jsonl ${P}_train_sweep.jsonl
cp $O/traffic.json profiles/r06_traffic.json
python - <<PY
import json
b = json.load(open("${P}_bench.json"))
print("bench:", round(b["value"], 3), "images/s;", round(b["unet_step_ms"], 3), "ms per UNet evaluation; p50", round(b["unet_step_ms_p50"], 3), "p90", round(b["unet_step_ms_p90"], 3))
r = b["roofline"]; print("roofline:", r["kernel"], round(r["achieved"], 1), r["unit"], "frac", round(r["frac"], 4), "traffic", r["traffic"], r["traffic_source"])
print("traffic stamp:", json.load(open("profiles/r06_traffic.json")).get("commit"))
print("sam:", json.load(open("${P}_sam_encoder.json")).get("latency_ms_p50"))
PY
grep -E "passed|failed" ${P}_pytest_gpu.txt | tail -1
