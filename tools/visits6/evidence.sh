#!/bin/bash
# Round 6 evidence visit: tools/evidence_round.sh (smoke, full -m gpu suite, bench + default bench, rocprofv3 stats, PMC families, traffic stamped with .commit_id,
# library table, training sweep) + the throttle probe beside the bench, the fused kernels' labs, the SAM encoder, and a PMC pass on the fused feed-forward kernel.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/evidence_round.sh
timeout 300 python tools/throttle_probe.py $OUT/throttle_bench_final.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline | cut -c1-900
timeout 200 python tools/ff_fused_lab.py --rounds 2 > $OUT/ff_fused_lab.json 2>/dev/null; cut -c1-900 $OUT/ff_fused_lab.json
AE_XATTN_FUSED=1 timeout 200 python tools/xattn_fused_lab.py --rounds 2 > $OUT/xattn_fused_lab.json 2>/dev/null; cut -c1-500 $OUT/xattn_fused_lab.json
bash tools/pmc.sh ff_a "ff_fused" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU > /dev/null 2>&1; echo "pmc ff rc=$?"
( timeout 300 python tools/bench_sam.py ) > $OUT/sam_encoder.json 2>/dev/null; tail -c 600 $OUT/sam_encoder.json
( timeout 300 python tools/gemm_vs_lib.py --sam ) > $OUT/sam_gemm_vs_library.txt 2>/dev/null; cat $OUT/sam_gemm_vs_library.txt
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sam -o sam -- python $R/tools/bench_sam.py --iters 10 ) > /dev/null 2>&1; cd $R
f=$(find $OUT/prof_sam -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/sam_kernel_stats.csv && head -8 $OUT/sam_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof_sam
