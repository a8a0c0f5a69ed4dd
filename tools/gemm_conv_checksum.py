"""Checksums of conv3x3 / GEMM outputs (every epilogue kind the UNet uses) on seeded inputs — for an A/B of two builds: equal checksums
on both sides = the builds compute the same bits.  AE_LIB_PATH selects the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)


def show(tag, t):
    f = t.float()
    print(f"{tag:44s} {float(f.sum()):.7e} {float(f.abs().sum()):.7e}")


for B, H, Cin, Cout, ups, stride in ((12, 64, 320, 320, False, 1), (12, 32, 640, 640, False, 1), (12, 16, 1280, 1280, True, 1), (12, 64, 320, 320, False, 2),
                                     (12, 8, 1280, 1280, False, 1), (2, 16, 2560, 1280, False, 1), (12, 64, 960, 320, False, 1), (12, 16, 1280, 1280, False, 1),
                                     (12, 16, 2560, 1280, False, 1), (12, 16, 640, 1280, False, 1), (3, 64, 320, 320, False, 1)):
    x = (r(B * H * H, Cin) * 0.5).to(BF).to(dev)
    w = (r(Cout, Cin, 3, 3) * (9 * Cin) ** -0.5)
    bias, add = r(Cout).to(dev), r(B, Cout).to(dev)
    Ho = H * 2 if ups else H // stride
    res = (r(B * Ho * Ho, Cout) * 0.5).to(BF).to(dev)
    ko = ops.conv_k_order(B * Ho * Ho, Cin, Cout, stride, ups)
    pk = ops.pack_conv3x3(w.to(dev), k_order=ko)
    cs = ops.colstats_buffer(B * Ho * Ho, Cout, dev)
    y, _, _ = ops.conv3x3(x, pk, bias, B, H, H, addvec=add, residual=res, stride=stride, upsample2x=ups, colstats=cs, k_order=ko)
    show(f"conv {B}x{H}x{H} {Cin}->{Cout} ups={int(ups)} s={stride}", y)
    show("   colstats", cs)
    y32, _, _ = ops.conv3x3(x, pk, bias, B, H, H, stride=stride, upsample2x=ups, out_f32=True, k_order=ko)
    show("   fp32 out", y32)
for M, N, K in ((49152, 320, 320), (12288, 640, 640), (12288, 5120, 640), (3072, 1280, 1280), (3072, 10240, 1280), (768, 1280, 1280), (936, 640, 768)):
    a = (r(M, K) * 0.5).to(BF).to(dev)
    w = (r(N, K) * K ** -0.5).to(BF).to(dev)
    bias = r(N).to(dev)
    res = (r(M, N) * 0.5).to(BF).to(dev)
    cs = ops.colstats_buffer(M, N, dev) if M % 32 == 0 else None
    show(f"gemm {M}x{N}x{K} bias+res", ops.gemm(a, ops.pack_linear(w), bias, residual=res, colstats=cs))
    if cs is not None:
        show("   colstats", cs)
    if N % 2 == 0 and N >= 2560:
        wg, bg = ops.pack_geglu(w, bias)
        show("   geglu", ops.gemm(a, wg, bg, epilogue=ops.EPI_GEGLU))
    show("   silu", ops.gemm(a, ops.pack_linear(w), bias, epilogue=ops.EPI_SILU))
