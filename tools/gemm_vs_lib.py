#!/usr/bin/env python3
"""Our dense GEMM kernels against the ROCm GEMM library (torch.matmul -> hipBLASLt / rocBLAS) on the UNet's plain linear layers
(no GEGLU): same operands, HIP-event timing, operands rotated through a pool larger than L2 + MALL so neither side sees them hot.
    python tools/gemm_vs_lib.py [--sam]     (--sam: the ViT-H image encoder's five GEMM shapes instead)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
SHAPES = [(12288, 640, 640, "to_out / proj L2"), (12288, 1920, 640, "qkv L2"), (12288, 640, 2560, "ff2 L2"), (12288, 640, 1280, "skip1x1 L2"),
          (3072, 1280, 1280, "to_out / proj L3"), (3072, 3840, 1280, "qkv L3"), (3072, 1280, 5120, "ff2 L3"), (3072, 1280, 2560, "skip1x1 L3"),
          (768, 1280, 1280, "to_out L4"), (768, 3840, 1280, "qkv L4"), (768, 1280, 5120, "ff2 L4"),
          (49152, 320, 320, "to_out L1 (row-panel)"), (49152, 960, 320, "qkv L1")]
SAM_SHAPES = [(4900, 3840, 1280, "SAM qkv (windows)"), (4900, 1280, 1280, "SAM proj (windows)"), (4096, 5120, 1280, "SAM lin1"),
              (4096, 1280, 5120, "SAM lin2"), (4096, 3840, 1280, "SAM qkv (global)")]


def timeit(fn, n_pool, iters=24, warm=4):
    for i in range(warm):
        fn(i % n_pool)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % n_pool)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    print(f"{'shape':44s} {'ours us':>9s} {'TF/s':>7s} {'lib us':>9s} {'TF/s':>7s}  lib/ours   (+bias+residual: ours / lib addmm-style)")
    for M, N, K, tag in (SAM_SHAPES if "--sam" in sys.argv else SHAPES):
        per = 2 * (M * K + N * K + 2 * M * N)
        n_pool = max(2, min(16, (600 << 20) // per))   # > 512 MB of distinct operands in rotation
        A = [torch.randn(M, K, device=DEV).to(BF) for _ in range(n_pool)]
        W = [torch.randn(N, K, device=DEV).to(BF) for _ in range(n_pool)]
        R = [torch.randn(M, N, device=DEV).to(BF) for _ in range(n_pool)]
        bias = torch.randn(N, device=DEV)
        bias_bf = bias.to(BF)
        out = torch.empty(M, N, device=DEV, dtype=BF)
        t_ours = timeit(lambda i: ops.gemm(A[i], W[i], out=out), n_pool)
        t_lib = timeit(lambda i: torch.matmul(A[i], W[i].t(), out=out), n_pool)
        t_ours_e = timeit(lambda i: ops.gemm(A[i], W[i], bias, residual=R[i], out=out), n_pool)
        t_lib_e = timeit(lambda i: torch.addmm(R[i], A[i], W[i].t(), out=out).add_(bias_bf), n_pool)
        fl = 2.0 * M * N * K
        print(f"{tag + f' {M}x{N}x{K}':44s} {t_ours:9.1f} {fl / t_ours / 1e6:7.0f} {t_lib:9.1f} {fl / t_lib / 1e6:7.0f}  {t_lib / t_ours:6.2f}     {t_ours_e:7.1f} / {t_lib_e:7.1f}", flush=True)


if __name__ == "__main__":
    main()
