#!/bin/bash
# round 4 visit 20: plan knobs re-measured in situ now that the 192x320 tile runs the ping-pong loop
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
run() { echo -n "$1: "; env $1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"; }
{ run "AE_X=0"; run "AE_GEMM_T320=15"; run "AE_CONV_T320_SPLITK=1"; run "AE_CONV_T320_SPLITK=3"; run "AE_X=0"; run "AE_GEMM_T320=15"; run "AE_CONV_T320_SPLITK=1"; run "AE_CONV_T320_SPLITK=3"; run "AE_GEMM_ROWPANEL=0"; } | tee gpurun_out/r04_v20_knobs.txt
