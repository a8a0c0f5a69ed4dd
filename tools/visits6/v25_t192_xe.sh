#!/bin/bash
# qkv of the 16x16 level (LayerNorm fold, 3072 x 3840 x 1280) on the 192x128 two-stage tile: AE_GEMM_T192_XE=0/1
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out/v25; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "layernorm_folded" 2>&1 | grep -E "passed|failed|error" | tail -3
python - <<'PY' 2>/dev/null | tee $OUT/kernel.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
from anyedit_amd import ops
M, C, N = 3072, 1280, 3840
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).bfloat16().cuda()
w, b = torch.randn(N, C, generator=g) / C ** 0.5, 0.1 * torch.randn(N, generator=g)
gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
wq, s, c = ops.pack_ln_fold(w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), geglu=False)
st = ops.rowstats_buffer(M, C, "cuda")
xs = x.float().reshape(M, C // 64, 64)
st[..., 0], st[..., 1] = xs.sum(-1), (xs * xs).sum(-1)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def run(): ops.gemm_ln(x, st, wq, s, c, 1e-5, out=out)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("AE_GEMM_T192_XE=%s: %.1f us" % (os.environ.get("AE_GEMM_T192_XE", "default"), e0.elapsed_time(e1) * 5))
PY
AE_GEMM_T192_XE=0 python - <<'PY' 2>/dev/null | tee -a $OUT/kernel.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
from anyedit_amd import ops
M, C, N = 3072, 1280, 3840
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).bfloat16().cuda()
w, b = torch.randn(N, C, generator=g) / C ** 0.5, 0.1 * torch.randn(N, generator=g)
gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
wq, s, c = ops.pack_ln_fold(w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), geglu=False)
st = ops.rowstats_buffer(M, C, "cuda")
xs = x.float().reshape(M, C // 64, 64)
st[..., 0], st[..., 1] = xs.sum(-1), (xs * xs).sum(-1)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def run(): ops.gemm_ln(x, st, wq, s, c, 1e-5, out=out)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("AE_GEMM_T192_XE=%s: %.1f us" % (os.environ.get("AE_GEMM_T192_XE", "default"), e0.elapsed_time(e1) * 5))
PY
for i in 1 2 3; do
  for f in 0 1; do
    echo "== AE_GEMM_T192_XE=$f (round $i)"
    AE_GEMM_T192_XE=$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  done
done | tee $OUT/ab.txt
