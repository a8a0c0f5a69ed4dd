import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def sub_sd(arrs, prefix, device="cpu", dtype=None):
    """Extract a state dict (torch tensors) from fixture arrays with the given key prefix."""
    out = {}
    for k, v in arrs.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.asarray(v)).to(device)
            if t.dtype == torch.int16:  # weights stored as bf16 bit patterns (the reference ran on the bf16-rounded values)
                t = t.view(torch.bfloat16).float()
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            out[k[len(prefix):]] = t
    return out


def T(a, device="cpu"):
    return torch.from_numpy(np.asarray(a)).to(device)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def psnr(a, b):
    """PSNR (dB) of a vs reference b with peak = dynamic range of b."""
    a = a.double()
    b = b.double()
    mse = float(((a - b) ** 2).mean())
    peak = float(b.max() - b.min())
    if mse == 0:
        return float("inf")
    return 10.0 * float(np.log10(peak * peak / mse))
