#!/bin/bash
# One GPU-box visit that collects the round's evidence: full -m gpu suite (with the printed parity numbers), the bench line, rocprofv3
# kernel stats of the same command, PMC passes on the attention kernel and the dominant conv, HBM traffic per kernel, the training
# sweep.  Logs -> gpurun_out/.  AE_EVIDENCE_CPU_E2E=1 adds BASELINE.md §3's end-to-end CPU legs to the bench line (4 more minutes).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
# AE_TEST_EDIT_CONTROL=1: the 96x96 5-step edit test runs its bf16-storage control for real (+5 minutes of host time)
( AE_TEST_EDIT_CONTROL=${AE_TEST_EDIT_CONTROL:-1} timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=8 ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rel-L2|slowest|s call" $OUT/pytest_gpu_full.log | tail -20
E2E=""; [ "${AE_EVIDENCE_CPU_E2E:-0}" = "1" ] && E2E="--cpu-e2e"
# power / clock samples while the bench runs (the driver's own smi.*.json are not visible to the builder)
( rocm-smi --showmaxpower --showpower --showclocks 2>&1 | head -60 ) > $OUT/smi_idle.txt
( while true; do date +%s.%N; rocm-smi --showpower --showclocks --showuse --json 2>/dev/null; sleep 0.2; done ) > $OUT/smi_bench.jsonl &
SMI=$!
( timeout 1500 python bench.py --steps 10 --warmup 2 --cpu-ops $E2E ) > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_full.json
( timeout 300 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"; cut -c1-300 $OUT/bench_default.json
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - $OUT/smi_bench.jsonl $OUT/smi_during_bench.json <<'PY'
import json, sys
pw, sclk, mclk, use = [], [], [], []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    for card, v in d.items():
        if not isinstance(v, dict):
            continue
        for k, x in v.items():
            try:
                if "Power" in k and "(W)" in k: pw.append(float(x))
                elif k.startswith("sclk clock speed"): sclk.append(float(str(x).strip("()Mhz")))
                elif k.startswith("mclk clock speed"): mclk.append(float(str(x).strip("()Mhz")))
                elif k == "GPU use (%)": use.append(float(x))
            except Exception:
                pass
def st(a):
    a = sorted(a)
    return {"n": len(a), "min": a[0], "p50": a[len(a) // 2], "max": a[-1]} if a else {"n": 0}
busy = [i for i, u in enumerate(use) if u >= 90]
out = {"what": "rocm-smi samples (5 Hz) while bench.py --steps 10 ran; 'busy' = samples with GPU use >= 90 %",
       "power_w": st(pw), "sclk_mhz": st(sclk), "mclk_mhz": st(mclk), "gpu_use_pct": st(use),
       "busy_power_w": st([pw[i] for i in busy if i < len(pw)]), "busy_sclk_mhz": st([sclk[i] for i in busy if i < len(sclk)])}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
cp $OUT/kernels_by_shape.json $OUT/kernels_by_shape_final.json 2>/dev/null
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/prof
bash tools/pmc.sh attn_a "attn self N=4096" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc attn rc=$?"
bash tools/pmc.sh conv_a "conv3x3 res" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc conv rc=$?"
bash tools/pmc.sh dense_a "gemm " SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc dense rc=$?"
bash tools/pmc.sh attnx_a "attn cross" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc attn cross rc=$?"
bash tools/pmc.sh gn_a "groupnorm" SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM > /dev/null 2>&1; echo "pmc gn rc=$?"
bash tools/traffic.sh > $OUT/traffic.log 2>&1; echo "traffic rc=$?"; tail -8 $OUT/traffic.log
( timeout 300 python tools/gemm_vs_lib.py ) > $OUT/gemm_vs_library.txt 2>/dev/null; echo "gemm_vs_lib rc=$?"; cat $OUT/gemm_vs_library.txt
rm -f $OUT/train_sweep.jsonl
for a in "" "--checkpoint" "--batch 16" "--batch 16 --checkpoint" "--batch 32 --checkpoint"; do timeout 200 python tools/bench_train.py --steps 10 --warmup 2 $a 2>/dev/null | tail -1 >> $OUT/train_sweep.jsonl; done
cut -c80-330 $OUT/train_sweep.jsonl
