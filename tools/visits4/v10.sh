#!/bin/bash
# round 4 visit 10: first run of the ping-pong self-attention kernel
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
{ AE_ATTN_PP=0 timeout 120 python tools/attn_pp_check.py; AE_ATTN_PP=1 timeout 120 python tools/attn_pp_check.py; AE_ATTN_PP=0 timeout 60 python tools/attn_pp_check.py | tail -1; AE_ATTN_PP=1 timeout 60 python tools/attn_pp_check.py | tail -1; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v10_attn_pp.txt
