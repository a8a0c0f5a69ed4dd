#!/bin/bash
# Round 5, visit 29: which limiter does the firmware report while the self-attention BACKWARD passes (d = 40, N = 4096) run alone?  (The four restructurings of
# v21-v23 and the in-wave pipeline of v28 all left its time unchanged.)
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 200 python tools/throttle_probe.py $OUT/v29_throttle_attn_bwd.json -- python tools/attn_bwd_lab.py 32 4096 40 8000 | cut -c1-1200
python -c "import json;d=json.load(open('$OUT/v29_throttle_attn_bwd.json'));print(d['workload_tail'][-300:].strip())"
