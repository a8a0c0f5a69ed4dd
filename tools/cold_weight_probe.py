#!/usr/bin/env python3
"""What do COLD weights cost a launch?  Each shape is timed with the same packed weight every call (weights L2 / MALL-hot, as in
tools/kbench.py) and with the weight rotated through a pool larger than the 256 MB MALL (every call streams its weights from HBM, as inside
a UNet evaluation, where 1.7 GB of weights pass between two uses of a layer).  Activations stay hot in both arms.
    python tools/cold_weight_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
B = 12


def timeit(fn, n, iters=60, warm=8):
    for i in range(warm):
        fn(i % n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    print(f"{'launch':52s} {'W MB':>7s} {'hot us':>8s} {'cold us':>8s}  cold/hot")
    for (H, Cin, Cout, tag) in ((64, 320, 320, "conv res L1"), (64, 960, 320, "conv res dec L1"), (32, 640, 640, "conv res L2"),
                                (32, 1920, 640, "conv res dec L2"), (16, 1280, 1280, "conv res L3"), (8, 1280, 1280, "conv res L4")):
        M = B * H * H
        ko = ops.conv_k_order(M, Cin, Cout)
        wbytes = Cout * 9 * Cin * 2
        n = max(2, min(256, (320 << 20) // wbytes))
        ws = [ops.pack_conv3x3(torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.02, k_order=ko) for _ in range(n)]
        x = torch.randn(M, Cin, device=DEV).to(BF)
        bias = torch.randn(Cout, device=DEV)
        hot = timeit(lambda i: ops.conv3x3(x, ws[0], bias, B, H, H, k_order=ko), n)
        cold = timeit(lambda i: ops.conv3x3(x, ws[i], bias, B, H, H, k_order=ko), n, iters=max(60, n))
        print(f"{tag + f' {Cin}->{Cout} @{H}':52s} {wbytes / 1e6:7.1f} {hot:8.1f} {cold:8.1f}  {cold / hot:6.2f}", flush=True)
    for (M, N, K, tag) in ((12288, 640, 640, "proj L2"), (12288, 1920, 640, "qkv L2"), (12288, 640, 2560, "ff2 L2"), (3072, 1280, 1280, "proj L3"),
                           (3072, 3840, 1280, "qkv L3"), (3072, 1280, 5120, "ff2 L3"), (768, 1280, 1280, "proj L4"), (49152, 320, 1280, "ff2 L1")):
        wbytes = N * K * 2
        n = max(2, min(256, (320 << 20) // wbytes))
        ws = [(torch.randn(N, K, device=DEV) * 0.02).to(BF) for _ in range(n)]
        a = torch.randn(M, K, device=DEV).to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        hot = timeit(lambda i: ops.gemm(a, ws[0], out=out), n)
        cold = timeit(lambda i: ops.gemm(a, ws[i], out=out), n, iters=max(60, n))
        print(f"{'gemm ' + tag + f' M={M} N={N} K={K}':52s} {wbytes / 1e6:7.1f} {hot:8.1f} {cold:8.1f}  {cold / hot:6.2f}", flush=True)


if __name__ == "__main__":
    main()
