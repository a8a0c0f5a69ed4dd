#!/usr/bin/env python3
"""The three small-M dense shapes that dominate the launch count of a batch-4 training step (tools/bench_train.py: ~235 launches of 16-18 us), under the
tile plans the library can be forced into: prints us per launch (HIP events, operands rotated through a pool larger than L2 + MALL).
    AE_GEMM_TILE=<0|1|2> AE_ROWPANEL_ANY_M=<0|1> python tools/train_gemm_plans.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

SHAPES = [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280), (16384, 960, 320), (256, 1280, 1280)]


def main():
    tag = f"AE_GEMM_TILE={os.environ.get('AE_GEMM_TILE', '-')} AE_ROWPANEL_ANY_M={os.environ.get('AE_ROWPANEL_ANY_M', '-')} AE_GEMM_DEEP64_MAX={os.environ.get('AE_GEMM_DEEP64_MAX', '-')}"
    out = []
    for M, N, K in SHAPES:
        per = 2 * (M * K + N * K + 2 * M * N)
        n_pool = max(4, min(64, (600 << 20) // per))
        A = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(n_pool)]
        W = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(n_pool)]
        R = [torch.randn(M, N, device="cuda").bfloat16() for _ in range(n_pool)]
        bias = torch.randn(N, device="cuda")
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for i in range(8):
            ops.gemm(A[i % n_pool], W[i % n_pool], bias, residual=R[i % n_pool], out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(64):
            ops.gemm(A[i % n_pool], W[i % n_pool], bias, residual=R[i % n_pool], out=o)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{M}x{N}x{K}: {e0.elapsed_time(e1) / 64 * 1e3:6.1f}")
    print(tag, "|", "  ".join(out))


if __name__ == "__main__":
    main()
