"""Ping-pong self-attention check (round 4): attn_pp_kernel against an fp32 torch statement of softmax(q k^T) v on the GPU (sampled heads), a
forced-rescale input, run-to-run bit equality, and the kernel time.  AE_ATTN_PP=0 runs the round-3 kernel through the same script."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(5)
BH, N, D = 96, 4096, 40
q, k, v = (torch.randn(BH, N, D, generator=g, device=dev).to(BF) for _ in range(3))
k[:, 1000] = q[:, 7] * 5.0      # a late spike: the lazy offset moves in some query groups only
k[:, 3000] = q[:, 600] * 8.0
out = ops.attention_bhnd(q, k, v).float()
worst = 0.0
for h in (0, 1, 47, 95):
    s = (q[h].float() @ k[h].float().T) * D ** -0.5
    ref = torch.softmax(s, -1) @ v[h].float()
    e = float((out[h] - ref).norm() / ref.norm())
    worst = max(worst, e)
print(f"rel-L2 vs fp32 torch (4 heads): {worst:.3e}", "OK" if worst < 6e-3 else "FAIL")
same = all(torch.equal(ops.attention_bhnd(q, k, v).float(), out) for _ in range(20))
print("20 repeated launches bit-identical:", same)
# ragged query count
out2 = ops.attention_bhnd(q[:, :N - 40].contiguous(), k, v).float()
print("ragged Nq: rows of full waves identical:", torch.equal(out2[:, :3840], out[:, :3840]), " tail rel-L2:",
      float((out2[:, 3840:] - out[:, 3840:N - 40]).norm() / out[:, 3840:N - 40].norm()))
q, k, v = (torch.randn(BH, N, D, generator=g, device=dev).to(BF) for _ in range(3))
for _ in range(3):
    ops.attention_bhnd(q, k, v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention_bhnd(q, k, v)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"AE_ATTN_PP={os.environ.get('AE_ATTN_PP', '1')}: {us:.1f} us  {4.0 * BH * N * N * D / us / 1e6:.1f} TFLOP/s (algorithmic, d = 40)")
