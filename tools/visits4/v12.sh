#!/bin/bash
# round 4 visit 12: ping-pong attention — which waves share a SIMD?  (group assignment variants; lock step as the control)
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
{ 
AE_ATTN_PP=0 timeout 60 python tools/attn_pp_check.py | tail -1
echo -n "group = wave >> 2: "; timeout 60 python tools/attn_pp_check.py | tail -1
for v in g1 g2 g3; do echo -n "variant $v: "; AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_$v.so timeout 60 python tools/attn_pp_check.py | tail -4 | tr '\n' ' '; echo; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_v12_attn_pp_grp.txt
