"""GroupNorm(+SiLU) launch times at the small feature maps (single-launch slab kernel), for an A/B of two builds on one box:
  AE_LIB_PATH=anyedit_amd/libanyedit_hip_prev.so python tools/gn_slab_probe.py ; python tools/gn_slab_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
for B, hw, C in ((12, 256, 1920), (12, 256, 1280), (12, 256, 2560), (12, 64, 1280), (12, 64, 2560), (24, 256, 1920), (4, 256, 1920)):
    x = torch.randn(B * hw, C, generator=g).to(torch.bfloat16).to(dev)
    ga, be = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    out = torch.empty_like(x)
    for _ in range(5):
        ops.groupnorm(x, ga, be, B, hw, 1e-5, silu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        ops.groupnorm(x, ga, be, B, hw, 1e-5, silu=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    xf = x.float().view(B, hw, 32, C // 32)
    mu = xf.mean((1, 3), keepdim=True)
    var = xf.var((1, 3), unbiased=False, keepdim=True)
    ref = torch.nn.functional.silu(((xf - mu) * torch.rsqrt(var + 1e-5)).view(B * hw, C) * ga + be)
    err = float((out.float() - ref).norm() / ref.norm())
    print(f"B={B:2d} HW={hw:4d} C={C:4d}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us per call (host-paced), rel-L2 vs fp32 {err:.2e}, "
          f"checksum {float(out.float().sum()):.6e}")
