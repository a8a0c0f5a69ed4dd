"""Seconds of back-to-back UNet self-attention launches (N = 4096, d = 40, B*H = 96) for tools/throttle_probe.py: which limiter does the firmware report while
the attention kernel alone runs, and at which clock?   python tools/throttle_probe.py OUT.json -- python tools/attn_burn.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
g = torch.Generator(device="cuda").manual_seed(3)
q, k, v = (torch.randn(96, 4096, 40, generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(5):
    ops.attention_bhnd(q, k, v)
torch.cuda.synchronize()
t0, n = time.time(), 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(200):
        ops.attention_bhnd(q, k, v)
    n += 200
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"attention N=4096 d=40 BH=96 (AE_ATTN_V={os.environ.get('AE_ATTN_V', 'default')}): {n} launches, {us:.1f} us each = {4.0 * 96 * 4096 * 4096 * 40 / us / 1e6:.1f} TFLOP/s")
