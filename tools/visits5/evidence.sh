#!/bin/bash
# Round 5 evidence visit: tools/evidence_round.sh (smoke, full -m gpu suite with the edit control for real, bench + default bench, rocprofv3 stats, PMC
# families, traffic, library table, training sweep) + the throttle probe beside the bench.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
bash tools/evidence_round.sh
timeout 300 python tools/throttle_probe.py $OUT/throttle_bench_final.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline | cut -c1-900
( timeout 300 python tools/bench_sam.py ) > $OUT/sam_encoder.json 2>/dev/null; tail -c 600 $OUT/sam_encoder.json
