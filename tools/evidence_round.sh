#!/bin/bash
# One GPU-box visit that collects the round's evidence: full -m gpu suite (with the printed parity numbers), the bench line with every
# CPU leg, rocprofv3 kernel stats of the same command, PMC passes on the attention kernel, HBM traffic per kernel.  Logs -> gpurun_out/
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
git -C $R rev-parse --short HEAD > $R/.commit_id 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=8 ) > $OUT/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rel-L2|slowest|s call" $OUT/pytest_gpu_full.log | tail -20
( timeout 900 python bench.py --steps 10 --warmup 2 --cpu-e2e --cpu-ops ) > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_full.json
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-160
bash tools/pmc.sh attn_a "attn self N=4096" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc a rc=$?"
bash tools/pmc.sh attn_b "attn self N=4096" GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > /dev/null 2>&1; echo "pmc b rc=$?"
bash tools/traffic.sh > $OUT/traffic.log 2>&1; echo "traffic rc=$?"; tail -8 $OUT/traffic.log
