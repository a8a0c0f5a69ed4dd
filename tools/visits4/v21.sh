#!/bin/bash
# round 4 visit 21: chunk-major K order on the split-K 192x320 convs too (AE_CONV_KMAJOR=2) under the ping-pong loop
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
run() { echo -n "$1: "; env $1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"; }
{ run "AE_CONV_KMAJOR=1"; run "AE_CONV_KMAJOR=2"; run "AE_CONV_KMAJOR=1"; run "AE_CONV_KMAJOR=2";
  for k in 1 2; do echo "== kbench AE_CONV_KMAJOR=$k"; AE_CONV_KMAJOR=$k timeout 100 python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L3|L1"; done; } | tee gpurun_out/r04_v21_kmajor_splitk.txt
