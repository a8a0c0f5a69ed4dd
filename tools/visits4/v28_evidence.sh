#!/bin/bash
# round 4 second evidence visit (LayerNorm fold + slab conv loop in): tools/evidence_round.sh with the real edit control, SAM encoder, launch trace
AE_TEST_EDIT_CONTROL=1 bash tools/evidence_round.sh
( timeout 200 python tools/bench_sam.py ) > gpurun_out/sam_encoder.json 2>/dev/null; tail -c 600 gpurun_out/sam_encoder.json
( timeout 200 python tools/trace_gaps.py ) > gpurun_out/trace_gaps.json 2>/dev/null; tail -c 400 gpurun_out/trace_gaps.json
