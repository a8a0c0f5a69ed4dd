#!/bin/bash
# Round 5 evidence visit: tools/evidence_round.sh (smoke, full -m gpu suite with the edit control for real, bench + default bench, rocprofv3 stats, PMC
# families, traffic, library table, training sweep) + the throttle probe beside the bench.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/evidence_round.sh
timeout 300 python tools/throttle_probe.py $OUT/throttle_bench_final.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline | cut -c1-900
B=$R/tools/ubench/build
timeout 120 python tools/throttle_probe.py $OUT/throttle_attn.json -- python tools/attn_burn.py 4 | cut -c1-700
AE_LAB_ITERS=150000 timeout 120 python tools/throttle_probe.py $OUT/throttle_gemm_proj.json -- $B/pp_plain p | cut -c1-700
AE_LAB_ITERS=40000 timeout 120 python tools/throttle_probe.py $OUT/throttle_gemm_geglu.json -- $B/pp_plain g | cut -c1-700
for f in attn gemm_proj gemm_geglu; do python -c "import json;d=json.load(open('$OUT/throttle_$f.json'));print('$f', d['workload_tail'][-200:].strip())"; done
( timeout 300 python tools/bench_sam.py ) > $OUT/sam_encoder.json 2>/dev/null; tail -c 600 $OUT/sam_encoder.json
