#!/bin/bash
# round 4 visit 14: ping-pong attention v2 (reads first, maxima in the MFMA phase, two register sets)
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
{
AE_ATTN_PP=0 timeout 60 python tools/attn_pp_check.py | tail -1
timeout 120 python tools/attn_pp_check.py
AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_apptr.so timeout 120 python tools/attn_pp_trace.py | grep -E "rc|tile 21 wave [04]|tile 22 wave [04]"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v14_attn_pp2.txt
