#!/bin/bash
# round 4 visit 11: ping-pong attention, s_setprio placement
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
{ for r in 1 2; do
AE_ATTN_PP=0 timeout 60 python tools/attn_pp_check.py | tail -1
echo -n "prio none: "; timeout 60 python tools/attn_pp_check.py | tail -1
echo -n "prio MFMA phase: "; AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_app1.so timeout 60 python tools/attn_pp_check.py | tail -1
echo -n "prio VALU phase: "; AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_app2.so timeout 60 python tools/attn_pp_check.py | tail -1
done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v11_attn_pp_prio.txt
