"""Measured ceilings of the box next to the datasheet ones: a library bf16 GEMM (hipBLASLt through
torch.matmul) and a streaming copy / triad.  Plumbing only — nothing here is on the product path.  Output: one JSON line."""
import json
import time

import torch


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda", 0)
    out = {"device": torch.cuda.get_device_name(0)}
    for n in (4096, 8192):
        a = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
        b = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
        t = timeit(lambda: torch.matmul(a, b))
        out[f"hipblaslt_bf16_gemm_{n}_tflops"] = 2.0 * n ** 3 / t / 1e12
    # the UNet's own shapes through the library, for a like-for-like line in DESIGN.md
    for (M, N, K) in ((49152, 2560, 320), (49152, 320, 320), (12288, 5120, 640), (3072, 10240, 1280), (3072, 1280, 5120)):
        a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
        t = timeit(lambda: torch.nn.functional.linear(a, w))
        out[f"hipblaslt_linear_M{M}_N{N}_K{K}_tflops"] = 2.0 * M * N * K / t / 1e12
    n = 1 << 29  # 512 Mi bf16 = 1 GiB per array: past the 256 MiB Infinity Cache
    x = torch.empty(n, device=dev, dtype=torch.bfloat16).normal_()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    out["copy_GBps"] = 2.0 * n * 2 / t / 1e9
    z = torch.empty_like(x)
    t = timeit(lambda: torch.add(x, y, alpha=2.0, out=z))
    out["triad_GBps"] = 3.0 * n * 2 / t / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
