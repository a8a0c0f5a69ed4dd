#!/usr/bin/env python3
"""SamPredictor latency on MI355X: build_sam (ViT-H) with random-init weights, a synthetic 512x768 uint8 image,
set_image (host resize + preprocess + image encoder) and predict_torch with box prompts as AnyEdit's mask tool issues them
(tools/tool.py:182, 227-237).  Parity against the oracle: tests/test_hip_bench_shapes.py::test_sam_predictor_end_to_end_vs_oracle.
    python tools/bench_sam_predict.py [--boxes 3] [--iters 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402
from anyedit_amd.segment_anything import SamPredictor, build_sam  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    torch.manual_seed(0)
    with torch.device("cuda"):
        sam = build_sam()
    with torch.no_grad():
        for n, p in sam.named_parameters():
            if "rel_pos" in n or "pos_embed" in n:
                p.normal_(0, 0.02)
    sam = sam.to("cuda").requires_grad_(False)
    sam.image_encoder.use_hip_graph = True
    pred = SamPredictor(sam)
    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, (512, 768, 3), dtype=np.uint8)
    g = torch.Generator().manual_seed(4)
    xy = torch.rand(a.boxes, 2, 2, generator=g).sort(dim=1).values
    boxes = (xy * torch.tensor([768.0, 512.0])).reshape(a.boxes, 4)
    tb = pred.transform.apply_boxes_torch(boxes, image.shape[:2]).cuda()

    out = {"what": "SamPredictor (ViT-H), 512x768 image", "boxes": a.boxes}
    out["set_image_ms"] = timed(lambda: pred.set_image(image), a.iters)
    resized = torch.as_tensor(pred.transform.apply_image(image), device="cuda").permute(2, 0, 1).contiguous()[None]
    out["set_torch_image_ms"] = timed(lambda: pred.set_torch_image(resized, image.shape[:2]), a.iters)
    out["predict_torch_ms"] = timed(lambda: pred.predict_torch(None, None, boxes=tb, multimask_output=False), a.iters)
    with ops.OpProfiler() as prof:
        masks, iou, low = pred.predict_torch(None, None, boxes=tb, multimask_output=False)
    summ = prof.summary()
    out["predict_torch_kernel_ms"] = sum(v["ms"] for v in summ.values())
    out["predict_torch_kernels"] = {k: {"calls": v["calls"], "ms": v["ms"]} for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    assert masks.shape == (a.boxes, 1, 512, 768) and masks.dtype == torch.bool and torch.isfinite(low).all()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
