#!/usr/bin/env python3
"""BASELINE.json configs[4] end to end on one MI355X (SURVEY.md §8d cfg 5): the local-edit path
    768x768 image -> [detector boxes] -> SAM ViT-H box prompts -> merged mask -> kl-f8 encode -> masked-latent AnySD denoising
    (96x96 latents, 3-branch CFG, 50 DDIM steps, blend per global_tool.py:183-184) -> kl-f8 decode,
every stage on the HIP path, random-init weights of the real geometries, synthetic image and boxes (GroundingDINO's backbone / text
tower are outside the scope: the detector is the callable boundary of anyedit_amd.tools.tool.maskgeneration).  Reports per-stage
latency; SAM attention runs in bf16 (fp8 deferred).
    python tools/run_local_edit.py [--ddim-steps 50] [--boxes 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from anyedit_amd.anysd.pipeline import EditPipeline  # noqa: E402
from anyedit_amd.ldm.models.autoencoder import AutoencoderKL  # noqa: E402
from anyedit_amd.segment_anything import SamPredictor, build_sam  # noqa: E402

KL_F8 = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
             attn_resolutions=[], dropout=0.0)


class Clock:
    def __init__(self):
        self.ms = {}

    def __call__(self, name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.ms[name] = 1e3 * (time.perf_counter() - t0)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--boxes", type=int, default=2)
    ap.add_argument("--size", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    S, L = a.size, a.size // 8
    torch.manual_seed(0)
    with torch.device(dev):
        sam = build_sam()
        vae = AutoencoderKL(ddconfig=KL_F8, embed_dim=4)
    with torch.no_grad():
        for n, p in sam.named_parameters():
            if "rel_pos" in n or "pos_embed" in n:
                p.normal_(0, 0.02)
    sam = sam.to(dev).requires_grad_(False)
    sam.image_encoder.use_hip_graph = True
    vae.eval().requires_grad_(False)
    unet, moe, sched = bench.build_model(dev)
    pipe = EditPipeline(moe, sched, use_graph=True)
    predictor = SamPredictor(sam)

    rng = np.random.default_rng(5)
    image = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
    g = torch.Generator().manual_seed(6)
    xy = torch.rand(a.boxes, 2, 2, generator=g).sort(dim=1).values
    boxes = (xy * S).reshape(a.boxes, 4)
    _, _, ehs, null, ref, code = bench.synthetic_inputs(1, dev, 0, L)
    x_T = torch.randn(1, 4, L, L, generator=g).to(dev)

    def run(clock):
        clock("sam_set_image", lambda: predictor.set_image(image))
        tb = predictor.transform.apply_boxes_torch(boxes, image.shape[:2]).to(dev)
        mask = clock("sam_predict_merged", lambda: predictor.predict_torch_merged(tb))                      # [1,1,S,S] bool
        img = torch.from_numpy(image).to(dev).permute(2, 0, 1)[None].float() / 127.5 - 1.0
        lat = clock("vae_encode", lambda: vae.encode(img).mode() * 0.18215)
        m_lat = F.interpolate(mask.float(), size=(L, L), mode="nearest")                                      # mask at latent size
        out = clock("denoise", lambda: pipe.edit(x_T, lat, ehs, null, ref, code, steps=a.ddim_steps, mask=m_lat, x0=lat))
        rgb = clock("vae_decode", lambda: vae.decode(out / 0.18215))
        return mask, out, rgb

    run(Clock())                       # warm-up: weight packing, graph captures
    clock = Clock()
    mask, out, rgb = run(clock)
    assert mask.shape == (1, 1, S, S) and mask.dtype == torch.bool
    assert out.shape == (1, 4, L, L) and torch.isfinite(out).all() and rgb.shape == (1, 3, S, S) and torch.isfinite(rgb).all()
    total = sum(clock.ms.values())
    print(json.dumps({"what": f"configs[4] local edit, {S}x{S}, {a.boxes} boxes, {a.ddim_steps} DDIM steps x 3 CFG branches, batch 1",
                      "stage_ms": {k: round(v, 2) for k, v in clock.ms.items()}, "total_ms": round(total, 1),
                      "images_per_s": 1e3 / total, "mask_coverage": float(mask.float().mean()),
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
