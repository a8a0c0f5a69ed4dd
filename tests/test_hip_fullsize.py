"""Full-size parity (`-m gpu`): the SD-1.5-shaped UNet of BASELINE.json configs[1] (64x64 latents) and configs[4] (96x96 latents =
768x768 images, N = 9216 self-attention tokens) against the fp32 oracle on shared weights, and a multi-step 3-branch CFG 7.5 / 1.5
masked edit at 96x96.  Reference path: ldm/modules/diffusionmodules/openaimodel.py:754-786, ldm/models/diffusion/ddim.py:122-251,
tools/global_tool.py:160-184.

Tolerance derivation.  The HIP path stores activations and weights in bf16 and accumulates in fp32; the reference
runs fp32 on the CPU (and fp16 autocast on its GPU path).  The CONTROL is the oracle itself with every stored activation and every
weight rounded to bf16 (`oracle.ldm_ref.bf16_storage`, fp32 arithmetic otherwise): its distance to the fp32 oracle is the error
intrinsic to bf16 storage of this graph.  The tests assert   err(HIP) <= 1.5 x err(control)   — i.e. the kernels add at most half
again on top of what the storage format itself costs — plus an absolute cap.
"""
import math
import os
import time

import pytest
import torch

from conftest import rel_l2, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"

SD15 = dict(image_size=64, in_channels=8, model_channels=320, out_channels=4, num_res_blocks=2,
            attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
            transformer_depth=1, context_dim=768, legacy=False)

_CACHE = {}


def _model():
    """SD-1.5-shaped UNet + AnySD MoE wrapper, random init with the zero-init layers re-initialised (G1), shared by the tests."""
    if "m" not in _CACHE:
        from util_models import unzero, G
        from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
        from anyedit_amd.anysd.model import MoE
        torch.manual_seed(0)
        unet = UNetModel(**SD15)
        unzero(unet, G(1), std=0.02)
        unet.eval().requires_grad_(False)
        torch.manual_seed(2)
        moe = MoE(unet, expert_num=11)
        moe.eval().requires_grad_(False)
        sd = {k: v.detach().float().clone() for k, v in moe.state_dict().items()}
        _CACHE["m"] = (moe.to(DEV), sd)
    return _CACHE["m"]


def _threads():
    torch.set_num_threads(min(os.cpu_count() or 8, 32))


@pytest.mark.parametrize("latent", [64, 96])
def test_unet_full_size_vs_oracle_with_bf16_control(latent):
    from oracle import ldm_ref as L
    _threads()
    moe, sd = _model()
    unet_sd = {k[5:]: v for k, v in sd.items() if k.startswith("unet.")}
    g = torch.Generator().manual_seed(11 + latent)
    x = torch.randn(1, 8, latent, latent, generator=g)
    t = torch.tensor([501], dtype=torch.long)
    ctx = torch.randn(1, 77, 768, generator=g)
    t0 = time.time()
    with torch.no_grad():
        ref = L.unet_forward(unet_sd, SD15, x, t, ctx)
        with L.bf16_storage():
            ctl = L.unet_forward(L.bf16_weights(unet_sd), SD15, x, t, ctx)
        got = moe.unet(x.to(DEV), t.to(DEV), context=ctx.to(DEV)).float().cpu()
    e_hip, e_ctl = rel_l2(got, ref), rel_l2(ctl, ref)
    print(f"\nUNet {latent}x{latent}: HIP rel-L2 {e_hip:.3e} ({psnr(got, ref):.1f} dB), bf16-storage control {e_ctl:.3e} ({psnr(ctl, ref):.1f} dB), "
          f"oracle time {time.time() - t0:.1f} s")
    assert math.isfinite(e_hip) and e_hip <= 1.5 * e_ctl, f"HIP {e_hip:.3e} vs control {e_ctl:.3e}"
    assert e_hip <= 2e-2 and psnr(got, ref) >= 45.0


def test_unet_full_size_run_to_run_determinism():
    """The SD-1.5-shaped UNet at the bench batch (12) gives bit-identical outputs over repeated evaluations: every kernel on the path
    sums in a fixed order, and no launch may depend on timing (see tests/test_hip_ops.py::test_attention_run_to_run_determinism_full_size)."""
    moe, _ = _model()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(12, 8, 64, 64, generator=g).to(DEV)
    t = torch.randint(0, 1000, (12,), generator=g).to(DEV)
    ctx = torch.randn(12, 77, 768, generator=g).to(DEV)
    with torch.no_grad():
        first = moe.unet(x, t, context=ctx)
        for i in range(5):
            again = moe.unet(x, t, context=ctx)
            assert torch.equal(again, first), f"evaluation {i + 1} differs from evaluation 0 in {int((again != first).sum())} elements"


def test_masked_edit_5_steps_cfg_full_size_96():
    """configs[4] geometry: B = 1 edit, 3 CFG branches (7.5 / 1.5), 5 DDIM steps, mask / x0 blend at 96x96 latents, AnySD routing
    and adapters on; HIP pipeline vs oracle.ddim_ref.ip2p_edit_loop on the same weights and the same blend noise."""
    from oracle import ddim_ref as D, schedule_ref as S, anysd_ref as A, ldm_ref as L
    from anyedit_amd.anysd.pipeline import EditPipeline
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM
    _threads()
    moe, sd = _model()
    unet_sd = {k[5:]: v for k, v in sd.items() if k.startswith("unet.")}
    prefixes = [n + ".attn2." for n, m in moe.unet.named_modules() if m.__class__.__name__ == "BasicTransformerBlock"]
    H = 96
    g = torch.Generator().manual_seed(41)
    x_T = torch.randn(1, 4, H, H, generator=g)
    img_lat = torch.randn(1, 4, H, H, generator=g) * 0.18215
    ehs, null = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    ref_emb = torch.randn(1, 257, 1280, generator=g)
    code = torch.tensor([2])
    mask = torch.zeros(1, 1, H, H)
    mask[:, :, 20:70, 30:90] = 1.0
    x0 = torch.randn(1, 4, H, H, generator=g) * 0.18215
    blend_noise = torch.randn(1, 4, H, H, generator=g)
    buffers = S.register_schedule("linear", 1000, 0.00085, 0.0120)
    ref3 = torch.cat([ref_emb, ref_emb, torch.zeros_like(ref_emb)])
    code3 = torch.cat([code] * 3)
    steps = 5

    def run_oracle(weights, unet_weights):
        def unet_fn(x_in, t, text_embedding):
            return A.moe_forward(unet_weights, SD15, weights, prefixes, x_in, t, text_embedding, ref3, code3)
        return D.ip2p_edit_loop(unet_fn, buffers, steps, x_T, img_lat, ehs, null, 7.5, 1.5, mask=mask, x0=x0, noise_for_blend=blend_noise)

    # The bf16-storage control loop is measured in the same run (round 6: until then the default bound was 1.5 x a control RECORDED in round 2 and the
    # control itself ran only under AE_TEST_EDIT_CONTROL=1 — a bound that does not move with the code is not derived).  It doubles the host time of this
    # test (~2.5 minutes more on 32 threads); AE_TEST_EDIT_CONTROL=0 falls back to the recorded figure for quick local runs.
    t0 = time.time()
    with torch.no_grad():
        ref = run_oracle(sd, unet_sd)
        ctl = None
        if os.environ.get("AE_TEST_EDIT_CONTROL", "1") != "0":
            with L.bf16_storage():
                sdb = L.bf16_weights(sd)
                ctl = run_oracle(sdb, {k[5:]: v for k, v in sdb.items() if k.startswith("unet.")})
    t_oracle = time.time() - t0
    sched = DDPM(moe.unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120).to(DEV)
    pipe = EditPipeline(moe, sched, use_graph=True)
    pipe.randn = lambda shape, device=None: blend_noise.to(device)
    out = pipe.edit(x_T.to(DEV), img_lat.to(DEV), ehs.to(DEV), null.to(DEV), ref_emb.to(DEV), code.to(DEV), steps=steps,
                    s_txt=7.5, s_img=1.5, mask=mask.to(DEV), x0=x0.to(DEV)).float().cpu()
    e_hip = rel_l2(out, ref)
    e_ctl = rel_l2(ctl, ref) if ctl is not None else 3.77e-2
    print(f"\n5-step CFG edit @96x96: HIP rel-L2 {e_hip:.3e} ({psnr(out, ref):.1f} dB), bf16-storage control {e_ctl:.3e} "
          f"({'measured' if ctl is not None else 'recorded'}), oracle time {t_oracle:.1f} s")
    # outside the mask the result is q_sample(x0) exactly as the reference blends it
    keep = (mask == 0).expand_as(out)
    assert rel_l2(out[keep], ref[keep]) <= 1e-5
    assert math.isfinite(e_hip) and e_hip <= 1.5 * e_ctl, f"HIP {e_hip:.3e} vs control {e_ctl:.3e}"
    assert e_hip <= 4e-2 and psnr(out, ref) >= 36.0


@pytest.mark.skipif(os.environ.get("AE_TEST_EDIT50") != "1", reason="~15 minutes of host oracle time: AE_TEST_EDIT50=1 (evidence visits; record in profiles/)")
def test_edit_50_steps_cfg_full_size_64_metric_length():
    """Parity at the METRIC's own length (BASELINE.json configs[1]; ldm/models/diffusion/ddim.py:122-251, tools/global_tool.py:160-184): 50 DDIM steps,
    3-branch CFG 7.5 / 1.5 at 64x64 latents, AnySD routing and adapters on.  The HIP pipeline runs the PRODUCTION plan — four copies of the edit, UNet batch
    12, captured graph, i.e. the tile plans, the fused feed-forward and the row-panel kernels the bench line is measured on — and its first image is compared
    with `oracle.ddim_ref.ip2p_edit_loop` at batch 1 on the same weights, and with the bf16-storage control of the same loop: err(HIP) <= 1.5 x err(control).
    The four copies must agree bit for bit (no kernel mixes samples)."""
    from oracle import ddim_ref as D, schedule_ref as S, anysd_ref as A, ldm_ref as L
    from anyedit_amd.anysd.pipeline import EditPipeline
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM
    _threads()
    moe, sd = _model()
    unet_sd = {k[5:]: v for k, v in sd.items() if k.startswith("unet.")}
    prefixes = [n + ".attn2." for n, m in moe.unet.named_modules() if m.__class__.__name__ == "BasicTransformerBlock"]
    H, steps, copies = 64, int(os.environ.get("AE_TEST_EDIT50_STEPS", "50")), 4
    g = torch.Generator().manual_seed(77)
    x_T = torch.randn(1, 4, H, H, generator=g)
    img_lat = torch.randn(1, 4, H, H, generator=g) * 0.18215
    ehs, null = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    ref_emb = torch.randn(1, 257, 1280, generator=g)
    code = torch.tensor([5])
    buffers = S.register_schedule("linear", 1000, 0.00085, 0.0120)
    ref3 = torch.cat([ref_emb, ref_emb, torch.zeros_like(ref_emb)])
    code3 = torch.cat([code] * 3)
    trace = {}

    def run_oracle(weights, unet_weights, tag):
        inter = []

        def unet_fn(x_in, t, text_embedding):
            inter.append(x_in[:1, :4].clone())          # the latent entering each step (branch 0 of the 3-branch batch)
            return A.moe_forward(unet_weights, SD15, weights, prefixes, x_in, t, text_embedding, ref3, code3)
        out = D.ip2p_edit_loop(unet_fn, buffers, steps, x_T, img_lat, ehs, null, 7.5, 1.5)
        trace[tag] = inter
        return out

    t0 = time.time()
    with torch.no_grad():
        ref = run_oracle(sd, unet_sd, "ref")
        with L.bf16_storage():
            sdb = L.bf16_weights(sd)
            ctl = run_oracle(sdb, {k[5:]: v for k, v in sdb.items() if k.startswith("unet.")}, "ctl")
    t_oracle = time.time() - t0
    sched = DDPM(moe.unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120).to(DEV)
    pipe = EditPipeline(moe, sched, use_graph=True)
    rep = lambda t: t.repeat(copies, *([1] * (t.dim() - 1))).to(DEV)  # noqa: E731
    out4 = pipe.edit(rep(x_T), rep(img_lat), rep(ehs), rep(null), rep(ref_emb), code.repeat(copies).to(DEV), steps=steps, s_txt=7.5, s_img=1.5).float().cpu()
    for i in range(1, copies):
        assert torch.equal(out4[i], out4[0]), f"copy {i} of the edit differs from copy 0"
    out = out4[:1]
    e_hip, e_ctl = rel_l2(out, ref), rel_l2(ctl, ref)
    # growth of the control's distance over the loop (the latent entering step i), for the record
    grow = [rel_l2(c, r) for c, r in zip(trace["ctl"], trace["ref"])]
    marks = {i: grow[i] for i in (1, 2, 5, 10, 20, 30, 40, len(grow) - 1) if i < len(grow)}
    nmse = float(((out - ref) ** 2).mean()) / float(ref.max() - ref.min()) ** 2
    print(f"\n{steps}-step CFG edit @64x64 (UNet batch 12 plan): HIP rel-L2 {e_hip:.3e} ({psnr(out, ref):.1f} dB, MSE / range^2 {nmse:.2e}), bf16-storage control "
          f"{e_ctl:.3e} ({psnr(ctl, ref):.1f} dB), ratio {e_hip / e_ctl:.2f}, oracle time {t_oracle:.0f} s; control rel-L2 of the latent entering step i: "
          + ", ".join(f"{i}: {v:.2e}" for i, v in marks.items()))
    assert math.isfinite(e_hip) and e_hip <= 1.5 * e_ctl, f"HIP {e_hip:.3e} vs control {e_ctl:.3e}"
    assert nmse <= 1e-3, "north_star: within 1e-3 PSNR-equivalent, read as MSE / range^2 <= 1e-3 (DESIGN.md §4)"
