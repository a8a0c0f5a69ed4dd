"""Self-attention BACKWARD alone (the training step's 64x64-level shape: B*H = 32 heads, N = 4096, d = 40; or argv: BH N D): forward once for the
log-sum-exp, then `reps` backward calls (delta pre-pass + dQ pass + dK/dV pass).  For rocprofv3 --pmc / --kernel-trace (tools/visits5/v20.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

BH, N, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 4096, 40)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
g = torch.Generator(device="cuda").manual_seed(3)
q, k, v, do = (torch.randn(BH, N, D, generator=g, device="cuda").to(torch.bfloat16) for _ in range(4))
lse = torch.empty(BH, 1, N, dtype=torch.float32, device="cuda")
st = (N * D, 0, D)
out = ops.attention(q, k, v, BH, 1, N, N, D, D ** -0.5, st, st, st, lse=lse)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
for _ in range(2):
    ops.attention_bwd(q, k, v, do, lse, BH, 1, N, N, D, D ** -0.5, st, st, st, dq, dk, dv, st, st, st, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.attention_bwd(q, k, v, do, lse, BH, 1, N, N, D, D ** -0.5, st, st, st, dq, dk, dv, st, st, st, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"attention backward BH={BH} N={N} d={D}: {us:.1f} us per call (delta + dQ + dK/dV) = {10.0 * BH * N * N * D / us / 1e6:.1f} TFLOP/s algorithmic (5 products)")
