#!/bin/bash
# round 4 evidence visit (tools/evidence_round.sh with the end-to-end CPU legs and the real edit control)
AE_EVIDENCE_CPU_E2E=1 AE_TEST_EDIT_CONTROL=1 bash tools/evidence_round.sh
( timeout 200 python tools/bench_sam.py ) > gpurun_out/sam_encoder.json 2>/dev/null; tail -c 600 gpurun_out/sam_encoder.json
( timeout 200 python tools/trace_gaps.py ) > gpurun_out/trace_gaps.json 2>/dev/null; tail -c 400 gpurun_out/trace_gaps.json
