#!/bin/bash
# Round 5, visit 9: which kernels are power-limited — throttle probe beside the self-attention kernel alone, one short-K dense GEMM alone, one GEGLU GEMM alone.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; B=$R/tools/ubench/build
timeout 120 python tools/throttle_probe.py $OUT/throttle_attn.json -- python tools/attn_burn.py 4 | cut -c1-500
AE_ATTN_V=3 timeout 120 python tools/throttle_probe.py $OUT/throttle_attn_v3.json -- python tools/attn_burn.py 4 | cut -c1-500
AE_LAB_ITERS=150000 timeout 120 python tools/throttle_probe.py $OUT/throttle_gemm_proj.json -- $B/pp_plain p | cut -c1-500
AE_LAB_ITERS=40000 timeout 120 python tools/throttle_probe.py $OUT/throttle_gemm_geglu.json -- $B/pp_plain g | cut -c1-500
for f in attn attn_v3 gemm_proj gemm_geglu; do python -c "import json;d=json.load(open('$OUT/throttle_$f.json'));print('$f', d['workload_tail'][-220:].strip().replace(chr(10),' | '), '| busy per_ppt', d['busy_per_percent']['per_ppt_pwr'])"; done
