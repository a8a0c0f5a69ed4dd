#!/usr/bin/env python3
"""Diagnostic (dev tool): run-to-run determinism and error pattern of the long-sequence attention variants (AE_ATTN_V)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops
DEV, BF = "cuda", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(21)
BH, N, D = 96, 4096, 40
qq = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
kk = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
vv = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
v3 = (vv.float() + 3.0).to(BF)
R = int(os.environ.get("DIAG_RUNS", "6"))
outs = [ops.attention_bhnd(qq, kk, v3).float() for _ in range(6)]
torch.cuda.synchronize()
nbad = 0
for i in range(1, R):
    o = outs[i] if i < 6 else ops.attention_bhnd(qq, kk, v3).float()
    diff = (o != outs[0])
    n = int(diff.sum())
    if n:
        nbad += 1
        idx = diff.nonzero()
        print(f"run {i} vs run 0: {n} differing elements, max abs {float((o - outs[0]).abs().max()):.4f}; heads {torch.unique(idx[:, 0]).tolist()[:8]} rows {torch.unique(idx[:, 1]).tolist()[:10]}.. cols {torch.unique(idx[:, 2]).tolist()[:12]}")
print(f"{nbad} of {R - 1} repeat runs differ from run 0")
if os.environ.get("DIAG_ONLY_DET"):
    sys.exit(0)
ref_heads = (0, 5, 50, 95)
for h in ref_heads:
    ref = torch.softmax(qq[h].float().cpu() @ kk[h].float().cpu().t() * D ** -0.5, -1) @ v3[h].float().cpu()   # plain fp32 reference
    for i in (0, 3):
        err = (outs[i][h].cpu() - ref).abs()
        bad = (err > 0.05).nonzero()
        print(f"head {h} run {i}: max err {float(err.max()):.4f}, elements > 0.05: {bad.shape[0]}", end="")
        if bad.shape[0]:
            rows = bad[:, 0]
            print(f"  rows mod 256 histogram (top): {torch.bincount(rows % 256, minlength=256).topk(6)}  rows//256: {torch.unique(rows // 256)[:12].tolist()} cols: {torch.unique(bad[:,1]).tolist()[:12]}")
        else:
            print()
