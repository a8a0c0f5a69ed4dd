#!/bin/bash
# Round 5, visit 6: narrow conv kernel timing against the implicit-GEMM tile; hoisted time embedding A/B with more steps.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "narrow" ) 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v6_narrow_timing.txt
import torch, sys
sys.path.insert(0, '.')
from anyedit_amd import ops
dev='cuda'; BF=torch.bfloat16
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
B,H,W,Cin,Cout=12,64,64,320,4
x=torch.randn(B*H*W,Cin,device=dev).to(BF); w=torch.randn(Cout,Cin,3,3,device=dev)*0.02; b=torch.randn(Cout,device=dev)
wn=ops.pack_conv3x3(w,cin_pad=Cin); wo=ops.pack_conv3x3(w)
print("narrow kernel   %.1f us" % timeit(lambda: ops.conv3x3_narrow(x,wn,b,B,H,W,out_f32=True)))
print("implicit GEMM   %.1f us" % timeit(lambda: ops.conv3x3(x,wo,b,B,H,W,out_f32=True)))
# cold-ish: rotate through 8 inputs (252 MB > L2)
xs=[torch.randn(B*H*W,Cin,device=dev).to(BF) for _ in range(8)]
i=[0]
def rot(fn):
    def f():
        i[0]=(i[0]+1)%8; return fn(xs[i[0]])
    return f
print("narrow, rotating inputs   %.1f us" % timeit(rot(lambda xx: ops.conv3x3_narrow(xx,wn,b,B,H,W,out_f32=True))))
print("implicit, rotating inputs %.1f us" % timeit(rot(lambda xx: ops.conv3x3(xx,wo,b,B,H,W,out_f32=True))))
PY
echo "== bench A/B (alternating), 12 steps: AE_HOIST_TEMB with the narrow kernel off"
for i in 1 2 3; do
  for v in 0 1; do
    AE_CONV_NARROW=0 AE_HOIST_TEMB=$v timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HOIST=$v', d['value'], d['ms_per_step'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v6_bench_ab.txt
