"""Attention launch times + output checksums at the UNet's and SAM's shapes, for an A/B of two builds on one box:
  AE_LIB_PATH=anyedit_amd/libanyedit_hip_prev.so python tools/attn_ab_probe.py ; python tools/attn_ab_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
#            name                  BH   Nq    Nk    D
SHAPES = [("self 64x64 d40",       96, 4096, 4096,  40), ("self 32x32 d80",  96, 1024, 1024,  80), ("self 16x16 d160", 96,  256,  256, 160),
          ("cross 64x64 d40 k78",  96, 4096,   78,  40), ("cross 32x32 d80", 96, 1024,   78,  80), ("SAM global d80",  16, 4096, 4096,  80)]
for name, BH, Nq, Nk, D in SHAPES:
    q = torch.randn(BH, Nq, D, generator=g).to(BF).to(dev)
    k = torch.randn(BH, Nk, D, generator=g).to(BF).to(dev)
    v = torch.randn(BH, Nk, D, generator=g).to(BF).to(dev)
    for _ in range(3):
        out = ops.attention_bhnd(q, k, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        out = ops.attention_bhnd(q, k, v)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 4.0 * BH * Nq * Nk * D
    o = out.float()
    print(f"{name:22s} {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s   checksum {float(o.sum()):.6e} {float(o.abs().sum()):.6e}")
