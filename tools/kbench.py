#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the SD-1.5 UNet shapes of configs[1] (UNet batch 12): HIP-event timing on the launch stream,
algorithmic FLOP / bytes (SURVEY.md §8d formulas).  Dev tool for the optimisation loop; run on the GPU box:
    python tools/kbench.py [filter-substring]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
B = 12


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def rnd(*shape):
    return torch.randn(*shape, device=DEV).to(BF)


def report(name, us, flops=0.0, nbytes=0.0):
    print(f"{name:58s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cases = []

    def case(name, fn, flops=0.0, nbytes=0.0):
        if flt in name:
            cases.append((name, fn, flops, nbytes))

    # ---- attention (fused-qkv strided layout as CrossAttention.rows uses it)
    for (N, C, d) in ((4096, 320, 40), (1024, 640, 80), (256, 1280, 160), (64, 1280, 160)):
        h = 8
        qkv = rnd(B * N, 3 * C)
        s = (N * 3 * C, d, 3 * C)
        case(f"attn self N={N} d={d} BH={B * h}",
             lambda qkv=qkv, s=s, N=N, C=C, d=d: ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B, h, N, N, d, d ** -0.5, s, s, s),
             4.0 * B * h * N * N * d, 2.0 * B * h * d * 4 * N)
        q = rnd(B * N, C)
        kv = rnd(B * 78, 2 * C)
        case(f"attn cross N={N} d={d} Nk=78",
             lambda q=q, kv=kv, N=N, C=C, d=d: ops.attention(q, kv, kv[:, C:], B, h, N, 78, d, d ** -0.5, (N * C, d, C), (78 * 2 * C, d, 2 * C), (78 * 2 * C, d, 2 * C)),
             4.0 * B * h * N * 78 * d, 2.0 * B * h * d * (2 * N + 2 * 78))
    # ---- dense GEMMs
    for (M, N, K, tag) in ((49152, 960, 320, "qkv L1"), (49152, 320, 320, "proj L1"), (49152, 2560, 320, "ff1 L1 (geglu)"),
                           (49152, 320, 1280, "ff2 L1"), (12288, 1920, 640, "qkv L2"), (12288, 5120, 640, "ff1 L2 (geglu)"),
                           (12288, 640, 2560, "ff2 L2"), (3072, 3840, 1280, "qkv L3"), (3072, 10240, 1280, "ff1 L3 (geglu)"),
                           (3072, 1280, 5120, "ff2 L3"), (936, 640, 768, "ctx kv L1"), (12, 1280, 1280, "time_embed"),
                           (49152, 320, 960, "skip1x1 960->320"),
                           # the latency-bound linear layers of the 32x32 / 16x16 / 8x8 levels
                           (12288, 640, 640, "proj L2"), (12288, 640, 1280, "skip1x1 L2"), (3072, 1280, 1280, "proj L3"),
                           (3072, 1280, 2560, "skip1x1 L3"), (768, 1280, 1280, "proj L4"), (768, 3840, 1280, "qkv L4"),
                           (768, 1280, 5120, "ff2 L4")):
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=DEV)
        epi = ops.EPI_GEGLU if "geglu" in tag else ops.EPI_NONE
        res = rnd(M, N) if tag.startswith(("proj", "ff2")) else None   # to_out / proj_out / ff2 add the residual stream in their epilogue
        case(f"gemm {tag} M={M} N={N} K={K}", lambda a=a, w=w, bias=bias, epi=epi, res=res: ops.gemm(a, w, bias, residual=res, epilogue=epi),
             2.0 * M * N * K, 2.0 * (M * K + N * K + M * (N // 2 if epi else N) + (M * N if res is not None else 0)))
    # ---- round 6: the fused feed-forward launch of the 64x64 level (LayerNorm -> GEGLU -> ff2 + residual [-> proj_out + residual])
    if B * 4096 >= 192 * 192:
        M, C, Hd = B * 4096, 320, 1280
        xf = rnd(M, C)
        w1p, b1p = ops.pack_geglu(torch.randn(2 * Hd, C, device=DEV) / C ** 0.5, 0.1 * torch.randn(2 * Hd, device=DEV))
        w2img = ops.pack_ff2_fused(torch.randn(C, Hd, device=DEV) / Hd ** 0.5)
        gam, bet, b2 = torch.ones(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        w3, r3 = rnd(C, C), rnd(M, C)
        case(f"ff_fused L1 M={M} H={Hd}", lambda: ops.ff_fused(xf, gam, bet, 1e-5, w1p, b1p, w2img, b2, residual=xf), 6.0 * M * C * Hd, 2.0 * (3 * M * C + 3 * C * Hd))
        case(f"ff_fused+proj_out L1 M={M} H={Hd}", lambda: ops.ff_fused(xf, gam, bet, 1e-5, w1p, b1p, w2img, b2, residual=xf, w3=w3, b3=b2, residual3=r3),
             6.0 * M * C * Hd + 2.0 * M * C * C, 2.0 * (4 * M * C + 3 * C * Hd + C * C))
    # ---- conv3x3
    for (H, Cin, Cout, stride, ups, tag) in ((64, 320, 320, 1, False, "res L1"), (64, 960, 320, 1, False, "res dec L1"),
                                             (32, 640, 640, 1, False, "res L2"), (32, 1920, 640, 1, False, "res dec L2"),
                                             (16, 1280, 1280, 1, False, "res L3"), (16, 2560, 1280, 1, False, "res dec L3"),
                                             (8, 1280, 1280, 1, False, "res L4"), (8, 2560, 1280, 1, False, "res dec L4"),
                                             (64, 320, 320, 2, False, "down L1"), (32, 640, 640, 1, True, "up ->64"),
                                             (64, 8, 320, 1, False, "stem"), (64, 320, 4, 1, False, "head")):
        x = rnd(B * H * H, Cin)
        Ho = (2 * H if ups else H) // stride
        ko = ops.conv_k_order(B * Ho * Ho, Cin, Cout, stride, ups)   # the K order the UNet module gives this launch
        w = ops.pack_conv3x3(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.02, k_order=ko)
        bias = torch.randn(Cout, device=DEV)
        case(f"conv3x3 {tag} {Cin}->{Cout} @{H} s{stride}{' up' if ups else ''}{' kmajor' if ko else ''}",
             lambda x=x, w=w, bias=bias, H=H, stride=stride, ups=ups, ko=ko: ops.conv3x3(x, w, bias, B, H, H, stride=stride, upsample2x=ups, k_order=ko),
             2.0 * B * Ho * Ho * Cout * 9 * Cin, 2.0 * (B * H * H * Cin + 9 * Cin * Cout + B * Ho * Ho * Cout))
    # ---- norms
    for (HW, C, C1) in ((4096, 320, None), (4096, 960, 640), (1024, 640, None), (256, 1280, None), (64, 2560, 1280)):
        x = rnd(B * HW, C)
        g, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        if C1:
            x1, x2 = x[:, :C1].contiguous(), x[:, C1:].contiguous()
            case(f"groupnorm+silu concat [{B},{HW},{C1}+{C - C1}]", lambda x1=x1, x2=x2, g=g, bt=bt, HW=HW: ops.groupnorm(x1, g, bt, B, HW, 1e-5, silu=True, x2=x2),
                 0.0, 2.0 * 2 * B * HW * C)
        else:
            case(f"groupnorm+silu [{B},{HW},{C}]", lambda x=x, g=g, bt=bt, HW=HW: ops.groupnorm(x, g, bt, B, HW, 1e-5, silu=True), 0.0, 2.0 * 2 * B * HW * C)
    for (M, C) in ((49152, 320), (12288, 640), (3072, 1280)):
        x = rnd(M, C)
        g, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        case(f"layernorm [{M},{C}]", lambda x=x, g=g, bt=bt: ops.layernorm(x, g, bt), 0.0, 2.0 * 2 * M * C)

    print(f"# kbench on {torch.cuda.get_device_name(0)}  AE_ATTN_QF40={os.environ.get('AE_ATTN_QF40', 'default')}")
    for name, fn, fl, nb in cases:
        try:
            report(name, timeit(fn), fl, nb)
        except Exception as e:  # keep going: this is a survey tool
            print(f"{name:58s} FAILED: {e}", flush=True)


if __name__ == "__main__":
    main()
