"""GPU parity tests: SAM ViT image-encoder path (row A10) against reference-derived goldens; AnySD router / adapters
(row A9, parity unpinned) against the oracle's restatement of OUR spec; the 3-branch edit pipeline against the oracle loop."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden, sub_sd, T, rel_l2, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def close(got, ref, rl2=2e-2, db=36.0, what=""):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    e, p = rel_l2(got, ref), psnr(got, ref)
    assert e <= rl2 and p >= db, f"{what}: rel_l2={e:.3e} (<= {rl2}), psnr={p:.1f} dB (>= {db})"


# ------------------------------------------------------------------------------------------------------------ SAM
def test_window_partition_bit_exact():
    from anyedit_amd import ops
    g = load_golden("sam_tiny")
    for key, ws, B, H, W in (("win", 4, 2, 10, 10), ("win64", 14, 1, 64, 64)):
        x = T(g[f"{key}.x"])
        C = x.shape[-1]
        Cp = (C + 7) // 8 * 8  # the kernel moves 16-byte chunks
        xp = torch.zeros(B, H, W, Cp)
        xp[..., :C] = x
        xq = xp.to(BF)
        win, pad = ops.window_partition(xq.reshape(B * H * W, Cp).to(DEV), B, H, W, ws)
        assert list(pad) == g[f"{key}.pad"].tolist()
        ref = T(g[f"{key}.w"]).to(BF).float()
        got = win.float().cpu().reshape(-1, ws, ws, Cp)[..., :C]
        assert torch.equal(got, ref)
        back = ops.window_unpartition(win, B, H, W, ws).float().cpu().reshape(B, H, W, Cp)
        assert torch.equal(back, xq.float())


def same_up_to_contraction(a, b):
    """Two kernels with the same source arithmetic: hipcc's fp-contract=fast picks the fused / unfused form per kernel, so a few elements per
    million (those where the affine part cancels) may differ by one bf16 rounding step; everything else — and every padding row — is identical."""
    a, b = a.float(), b.float()
    d = (a - b).abs()
    nd = int((d > 0).sum())
    # one bf16 step at the larger magnitude, plus the fp32 rounding of the O(1) addends where x_hat * gamma and beta cancel
    step = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-6
    return nd <= max(2, a.numel() // 100000) and bool((d <= step).all()) and bool(((a == 0) == (b == 0)).all())


def test_layernorm_with_window_addressing():
    """ae_layernorm_window_bf16 (norm1 + partition; un-partition + shortcut + norm2: image_encoder.py:166-181) against the launches it
    replaces (same values up to the compiler's contraction choice, the residual sum bit for bit) and against the fp64 statement of the same
    lines on the bf16 inputs."""
    from anyedit_amd import ops
    g = torch.Generator().manual_seed(5)
    for (B, H, W, C, ws) in ((1, 64, 64, 1280, 14), (2, 10, 13, 320, 4), (1, 14, 14, 640, 14), (3, 5, 9, 2560, 7), (1, 33, 2, 1280, 8)):
        assert ops.layernorm_window_ok(C)
        x = (torch.randn(B * H * W, C, generator=g) * 2 + 0.3).to(BF).to(DEV)
        gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
        win, pad = ops.layernorm_window_partition(x, gamma, beta, 1e-6, B, H, W, ws)
        ref, pad2 = ops.window_partition(ops.layernorm(x, gamma, beta, 1e-6), B, H, W, ws)
        assert pad == pad2 and same_up_to_contraction(win, ref), (B, H, W, C, ws)
        xr = x.double().cpu().reshape(B, H, W, C)
        ln = torch.nn.functional.layer_norm(xr, (C,), gamma.double().cpu(), beta.double().cpu(), 1e-6)
        lnp = torch.nn.functional.pad(ln, (0, 0, 0, pad[1] - W, 0, pad[0] - H))
        lnw = lnp.reshape(B, pad[0] // ws, ws, pad[1] // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
        assert rel_l2(win.double().cpu(), lnw) < 4e-3
        npad = int((lnw.abs().sum(-1) == 0).sum())
        assert npad == B * (pad[0] * pad[1] - H * W) and int((win.float().abs().sum(-1) == 0).sum()) == npad
        a = torch.randn(win.shape[0], C, generator=g).to(BF).to(DEV)   # the attention output in window order
        xs, h2 = ops.window_merge_layernorm(a, x, gamma, beta, 1e-6, B, H, W, ws)
        xs_ref = ops.add_bcast(ops.window_unpartition(a, B, H, W, ws), x)
        assert torch.equal(xs, xs_ref) and same_up_to_contraction(h2, ops.layernorm(xs_ref, gamma, beta, 1e-6)), (B, H, W, C, ws)
    assert not ops.layernorm_window_ok(256)
    with pytest.raises(RuntimeError):   # outside the envelope: an error, not a silent other path
        ops.layernorm_window_partition(torch.zeros(16, 256, dtype=BF, device=DEV), torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 1e-6, 1, 4, 4, 2)


def test_relpos_terms_and_biased_attention():
    from anyedit_amd import ops
    from oracle import sam_ref as M, ldm_ref as L
    g = torch.Generator().manual_seed(21)
    for (B, heads, H, W, d) in ((2, 2, 8, 8, 32), (3, 2, 14, 14, 80), (1, 2, 64, 64, 80), (2, 2, 3, 64, 32), (1, 3, 5, 64, 40)):
        N = H * W
        q, k, v = (torch.randn(B * heads, N, d, generator=g).to(BF).float() for _ in range(3))
        rph, rpw = torch.randn(2 * H - 1, d, generator=g) * 0.3, torch.randn(2 * W - 1, d, generator=g) * 0.3
        rel_h, rel_w = M.decomposed_rel_pos_terms(q, rph, rpw, (H, W), (H, W))
        Rh, Rw = M.get_rel_pos(H, H, rph).contiguous().to(DEV), M.get_rel_pos(W, W, rpw).contiguous().to(DEV)
        qd = q.to(DEV, BF)
        gh, gw = ops.sam_relpos_terms(qd, (N * d, 0, d), Rh, Rw, B * heads, 1, H, W, d)
        assert rel_l2(gh.cpu().reshape(rel_h.shape), rel_h) < 1e-5 and rel_l2(gw.cpu().reshape(rel_w.shape), rel_w) < 1e-5
        bias = (rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).reshape(B * heads, N, N)
        ref = L.sdpa_core(q, k, v, d ** -0.5, bias=bias)
        out = ops.attention(qd, k.to(DEV, BF), v.to(DEV, BF), B * heads, 1, N, N, d, d ** -0.5, (N * d, 0, d), (N * d, 0, d),
                            (N * d, 0, d), rel_h=gh, rel_w=gw, kH=H, kW=W)
        e = rel_l2(out.float().cpu().reshape(ref.shape), ref)
        assert e < 6e-3, (H, W, d, e)


def test_relpos_terms_wave_groupings():
    """ae_sam_relpos_terms' matrix-pipe kernel (head dims 16 k; the line kernels otherwise): ragged tiles of (batch, head, query) triples, ragged
    line lengths, and the (batch, head) strides of a fused qkv row (image_encoder.py:227, 349-355) against the oracle's einsum."""
    from anyedit_amd import ops
    from oracle import sam_ref as M
    g = torch.Generator().manual_seed(22)
    for (B, heads, H, W, d) in ((1, 1, 5, 7, 32), (2, 3, 6, 5, 40), (4, 4, 7, 9, 64), (13, 4, 14, 14, 80), (1, 16, 9, 20, 80), (2, 40, 3, 2, 64)):
        N, C = H * W, heads * d
        qkv = torch.randn(B, N, 3, heads, d, generator=g).to(BF)
        q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(B * heads, N, d).float()
        rph, rpw = torch.randn(2 * H - 1, d, generator=g) * 0.3, torch.randn(2 * W - 1, d, generator=g) * 0.3
        rel_h, rel_w = M.decomposed_rel_pos_terms(q, rph, rpw, (H, W), (H, W))
        Rh, Rw = M.get_rel_pos(H, H, rph).contiguous().to(DEV), M.get_rel_pos(W, W, rpw).contiguous().to(DEV)
        gh, gw = ops.sam_relpos_terms(qkv.to(DEV).reshape(B * N, 3 * C), (N * 3 * C, d, 3 * C), Rh, Rw, B, heads, H, W, d)
        eh, ew = rel_l2(gh.cpu().reshape(rel_h.shape), rel_h), rel_l2(gw.cpu().reshape(rel_w.shape), rel_w)
        assert eh < 1e-5 and ew < 1e-5, (B, heads, H, W, d, eh, ew)


def test_sam_modules_golden():
    from anyedit_amd.segment_anything.modeling import image_encoder as E
    g = load_golden("sam_tiny")
    for tag in ("attn_g8", "attn_w14"):
        B, H, W, dim, heads = (int(v) for v in g[f"{tag}.cfg"])
        a = E.Attention(dim, num_heads=heads, qkv_bias=True, use_rel_pos=True, input_size=(H, W))
        a.load_state_dict(sub_sd(g, f"{tag}.w."))
        close(a.to(DEV)(T(g[f"{tag}.x"]).to(DEV)), g[f"{tag}.y"], what=tag)
    for tag, ws in (("blk_win", 4), ("blk_glob", 0)):
        b = E.Block(64, 2, use_rel_pos=True, window_size=ws, input_size=(10, 10), norm_layer=lambda d: nn.LayerNorm(d, eps=1e-6))
        b.load_state_dict(sub_sd(g, f"{tag}.w."))
        close(b.to(DEV)(T(g[f"{tag}.x"]).to(DEV)), g[f"{tag}.y"], what=tag)
    enc = E.ImageEncoderViT(img_size=80, patch_size=8, in_chans=3, embed_dim=64, depth=2, num_heads=2, mlp_ratio=4.0, out_chans=32,
                            qkv_bias=True, norm_layer=lambda d: nn.LayerNorm(d, eps=1e-6), use_abs_pos=True, use_rel_pos=True,
                            window_size=4, global_attn_indexes=(1,))
    enc.load_state_dict(sub_sd(g, "enc.w."))
    close(enc.to(DEV)(T(g["enc.x"]).to(DEV)), g["enc.y"], rl2=3e-2, db=32.0, what="ImageEncoderViT tiny")


def test_sam_vit_h_shapes_run():
    """ViT-H geometry (25 windows x 196 tokens, 4 global blocks of 4096 tokens) at depth 2 (1 windowed + 1 global)."""
    from functools import partial
    from anyedit_amd.segment_anything.modeling.image_encoder import ImageEncoderViT
    torch.manual_seed(0)
    with torch.device(DEV):
        enc = ImageEncoderViT(depth=2, embed_dim=1280, img_size=1024, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                              num_heads=16, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=(1,),
                              window_size=14, out_chans=256)
    x = torch.randn(1, 3, 1024, 1024, device=DEV)
    with torch.no_grad():
        y = enc(x)
    assert y.shape == (1, 256, 64, 64) and torch.isfinite(y).all()


# ------------------------------------------------------------------------------------------------------------ AnySD
def _tiny_moe():
    from util_models import build_tiny_unet, TINY_UNET
    from anyedit_amd.anysd.model import MoE
    g = load_golden("unet_tiny")
    unet = build_tiny_unet()
    unet.load_state_dict(sub_sd(g, "w."))
    torch.manual_seed(5)
    moe = MoE(unet, expert_num=11, n_tasks=6, context_dim=16, clip_dim=32, ip_tokens=4)
    with torch.no_grad():
        moe.task_embs.mul_(20.0)  # make the router non-trivial
        moe.gate.weight.mul_(8.0)
    return moe, TINY_UNET


def test_anysd_moe_forward_self_consistency():
    from oracle import anysd_ref as A
    moe, cfg = _tiny_moe()
    g = torch.Generator().manual_seed(31)
    B = 3
    x = torch.randn(B, 8, 8, 8, generator=g)
    t = torch.tensor([981, 501, 21])
    ehs = torch.randn(B, 5, 16, generator=g)
    ref_emb = torch.randn(B, 9, 32, generator=g)
    code = torch.tensor([0, 4, 2])
    sd = {k: v.detach().float() for k, v in moe.state_dict().items()}
    unet_sd = {k[5:]: v for k, v in sd.items() if k.startswith("unet.")}
    prefixes = [n + ".attn2." for n, m in moe.unet.named_modules() if m.__class__.__name__ == "BasicTransformerBlock"]
    with torch.no_grad():
        ref = A.moe_forward(unet_sd, cfg, sd, prefixes, x, t, ehs, ref_emb, code)
        probs_ref, top1_ref, _ = A.task_gate(sd["task_embs"], code, sd["gate.weight"], sd["gate.bias"])
        moe = moe.to(DEV)
        probs, top1, _ = moe.route(code.to(DEV))
        assert torch.equal(top1.cpu().long(), top1_ref) and torch.allclose(probs.cpu(), probs_ref, atol=1e-5)
        got = moe(x.to(DEV), t.to(DEV), ehs.to(DEV), ref_emb.to(DEV), code.to(DEV))
    close(got, ref, what="AnySD MoE forward (our spec)")
    # the adapters actually contribute: switching them off changes the output
    with torch.no_grad():
        ctx_rows, cache = moe.prepare_conditioning(ehs.to(DEV), ref_emb.to(DEV), code.to(DEV))
        cache = {k: v for k, v in cache.items() if not (isinstance(k, tuple) and k[0] == "adapter")}
        no_ad = moe.denoise(x.to(DEV), t.to(DEV), ctx_rows, cache)
    assert rel_l2(no_ad.cpu(), got.cpu()) > 1e-3


def test_edit_pipeline_vs_oracle_loop_and_graph_replay():
    """3-branch CFG + DDIM update + masked blend: HIP pipeline (eager and HIP-graph) vs the oracle loop on the same weights."""
    from oracle import ddim_ref as D, schedule_ref as S, anysd_ref as A
    from anyedit_amd.anysd.pipeline import EditPipeline
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM
    moe, cfg = _tiny_moe()
    g = torch.Generator().manual_seed(41)
    B = 2
    x_T = torch.randn(B, 4, 8, 8, generator=g)
    img_lat = torch.randn(B, 4, 8, 8, generator=g) * 0.18215
    ehs, null = torch.randn(B, 5, 16, generator=g), torch.randn(1, 5, 16, generator=g)
    ref_emb = torch.randn(B, 9, 32, generator=g)
    code = torch.tensor([1, 3])
    mask = (torch.rand(B, 1, 8, 8, generator=g) > 0.4).float()
    x0 = torch.randn(B, 4, 8, 8, generator=g)
    blend_noise = torch.randn(B, 4, 8, 8, generator=g)
    sd = {k: v.detach().float() for k, v in moe.state_dict().items()}
    unet_sd = {k[5:]: v for k, v in sd.items() if k.startswith("unet.")}
    prefixes = [n + ".attn2." for n, m in moe.unet.named_modules() if m.__class__.__name__ == "BasicTransformerBlock"]
    buffers = S.register_schedule("linear", 1000, 0.00085, 0.0120)
    ref3 = torch.cat([ref_emb, ref_emb, torch.zeros_like(ref_emb)])
    code3 = torch.cat([code] * 3)

    def unet_fn(x_in, t, text_embedding):
        return A.moe_forward(unet_sd, cfg, sd, prefixes, x_in, t, text_embedding, ref3, code3)

    steps = 5
    from oracle import ldm_ref as L
    with torch.no_grad():
        ref = D.ip2p_edit_loop(unet_fn, buffers, steps, x_T, img_lat, ehs, null.expand(B, -1, -1), 7.5, 1.5, mask=mask, x0=x0,
                               noise_for_blend=blend_noise)
        # control: the same loop with bf16 storage of activations / weights, fp32 arithmetic (the derived tolerance)
        sdb = L.bf16_weights(sd)
        usdb = {k[5:]: v for k, v in sdb.items() if k.startswith("unet.")}
        with L.bf16_storage():
            ctl = D.ip2p_edit_loop(lambda x_in, t, te: A.moe_forward(usdb, cfg, sdb, prefixes, x_in, t, te, ref3, code3), buffers, steps, x_T,
                                   img_lat, ehs, null.expand(B, -1, -1), 7.5, 1.5, mask=mask, x0=x0, noise_for_blend=blend_noise)
    e_ctl = rel_l2(ctl, ref)
    moe = moe.to(DEV)
    sched = DDPM(moe.unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120).to(DEV)
    outs = []
    for use_graph in (False, True):
        pipe = EditPipeline(moe, sched, use_graph=use_graph)
        pipe.randn = lambda shape, device=None: blend_noise.to(device)
        out = pipe.edit(x_T.to(DEV), img_lat.to(DEV), ehs.to(DEV), null.to(DEV), ref_emb.to(DEV), code.to(DEV), steps=steps,
                        s_txt=7.5, s_img=1.5, mask=mask.to(DEV), x0=x0.to(DEV))
        outs.append(out.cpu())
        assert np.array_equal(pipe.sampler.ddim_timesteps, S.make_ddim_timesteps("uniform", steps, 1000))
        close(out, ref, rl2=8e-2, db=30.0, what=f"edit pipeline (graph={use_graph})")   # absolute cap (30 dB = north_star's bound)
        e_hip = rel_l2(out.float().cpu(), ref)
        assert e_hip <= 1.5 * e_ctl + 1e-3, f"edit pipeline: HIP {e_hip:.3e} vs bf16-storage control {e_ctl:.3e}"
    assert torch.equal(outs[0], outs[1]), "HIP-graph replay must reproduce the eager launches bit for bit"
    # the time-embedding chain hoisted out of the loop (one M = steps pass, a row broadcast per step) against the per-step chain (AE_HOIST_TEMB=0): the
    # same GEMMs on the same rows -> the same latents; and `set_step` alone (what bench.py's roofline pass and the tools use) reproduces
    # the embedding rows the loop would have installed for that timestep
    import os
    os.environ["AE_HOIST_TEMB"] = "0"
    try:
        pipe0 = EditPipeline(moe, sched, use_graph=False)
        pipe0.randn = lambda shape, device=None: blend_noise.to(device)
        out0 = pipe0.edit(x_T.to(DEV), img_lat.to(DEV), ehs.to(DEV), null.to(DEV), ref_emb.to(DEV), code.to(DEV), steps=steps,
                          s_txt=7.5, s_img=1.5, mask=mask.to(DEV), x0=x0.to(DEV))
        assert pipe0._emb_pack is None
    finally:
        del os.environ["AE_HOIST_TEMB"]
    e_h = rel_l2(out0.float().cpu(), outs[0].float())
    assert e_h <= 2e-3, f"hoisted vs per-step time embedding: {e_h:.3e}"
    assert pipe._emb_pack is not None
    last = pipe._emb_pack.all.clone()
    pipe.set_step(int(np.flip(pipe.sampler.ddim_timesteps)[1]))
    assert not torch.equal(pipe._emb_pack.all, last)
    pipe.set_step(int(np.flip(pipe.sampler.ddim_timesteps)[-1]))
    assert torch.equal(pipe._emb_pack.all, last), "set_step must install the rows the loop installs for the same timestep"


# ------------------------------------------------------------------------------------------------------------ training step (A11)
def test_training_step_gradients_vs_oracle_autograd():
    """Row A11: eps-MSE loss and the gradients of EVERY trainable (image projection, per-expert adapter K/V projections, task
    embeddings) from the HIP forward + tape backward, against torch.autograd of the oracle's fp32 restatement of the same spec."""
    from oracle import anysd_ref as A, ddim_ref as D
    from anyedit_amd.anysd.train import AnySDTrainer
    moe, cfg = _tiny_moe()
    g = torch.Generator().manual_seed(77)
    B = 4
    lat = torch.randn(B, 4, 8, 8, generator=g)
    img = torch.randn(B, 4, 8, 8, generator=g) * 0.5
    noise = torch.randn(B, 4, 8, 8, generator=g)
    t = torch.tensor([981, 501, 21, 333])
    ehs = torch.randn(B, 5, 16, generator=g)
    ref_emb = torch.randn(B, 9, 32, generator=g)
    code = torch.tensor([0, 4, 2, 4])
    acp = torch.linspace(0.9999, 0.005, 1000)
    sa, s1 = acp.sqrt(), (1 - acp).sqrt()

    # oracle: fp32 autograd of our spec, and the bf16-storage CONTROL of the same graph (forward and backward storage rounded to bf16,
    # exact arithmetic): the tolerance of every gradient below is derived from it, as the forward tests do
    from util_models import oracle_training_grads, grad_tolerance, pooled_rel_l2, ROUTER_PATH
    sd = {k: v.detach().float().clone() for k, v in moe.state_dict().items()}
    prefixes = [n + ".attn2." for n, m in moe.unet.named_modules() if m.__class__.__name__ == "BasicTransformerBlock"]
    batch = (lat, img, noise, t, ehs, ref_emb, code, sa, s1)
    loss_ref_v, g_ref = oracle_training_grads(sd, cfg, prefixes, batch, control=False)
    loss_ctl_v, g_ctl = oracle_training_grads(sd, cfg, prefixes, batch, control=True)
    names = list(g_ref)

    moe = moe.to(DEV)
    tr = AnySDTrainer(moe, sa.to(DEV), s1.to(DEV), lr=1e-3)
    loss, tape, leaves = tr.forward_loss(lat.to(DEV), img.to(DEV), ehs.to(DEV), ref_emb.to(DEV), code.to(DEV), noise.to(DEV), t.to(DEV))
    assert abs(float(loss) - loss_ref_v) <= 1.5 * abs(loss_ctl_v - loss_ref_v) + 2e-3 * abs(loss_ref_v), (float(loss), loss_ref_v, loss_ctl_v)
    grads = tr.backward(tape, leaves)
    assert set(grads) == set(names)
    worst = 0.0
    print()
    for k in names:
        gr = g_ref[k]
        if gr is None or float(gr.abs().max()) == 0.0:
            assert float(grads[k].abs().max()) == 0.0, f"{k}: expected an all-zero gradient"
            continue
        e, e_ctl = rel_l2(grads[k].cpu(), gr), rel_l2(g_ctl[k], gr)
        worst = max(worst, e)
        # err(HIP) <= 1.5 x err(control) (2.5 x for gradients with a handful of non-zero entries: util_models.grad_tolerance).  The
        # router-gate path (task_embs, gate.*) is a cancelling inner product of bf16 gradients, so ITS control error is large too
        # (gate.bias: 0.2 for the exact-arithmetic control) — the bound follows the control instead of a hand-picked constant.
        tol = grad_tolerance(k, gr, e_ctl)
        print(f"  grad {k:34s} HIP {e:.3e}  control {e_ctl:.3e}  bound {tol:.3e}" + ("  (single batch: informational, pooled below)" if k in ROUTER_PATH else ""))
        if k not in ROUTER_PATH:
            assert e <= tol, f"grad {k}: HIP rel_l2 {e:.3e} vs bf16-storage control {e_ctl:.3e}"
        else:
            assert e <= 4.0 * e_ctl + 1e-3, f"grad {k}: HIP rel_l2 {e:.3e} vs bf16-storage control {e_ctl:.3e} (gross-error cap of the single batch)"
    assert worst > 0.0
    # Router path (gate.weight, gate.bias, task_embs), VERDICT r4: each of these gradients is a sum of B = 4 rank-one terms of cancelling bf16
    # gradients — one batch gives a noise norm that moves by 30 % under an fp32-ulp change of the forward.  Pool the error over NB independent
    # batches (fresh latents, noise, context, image embeddings, edit codes and time steps; same weights — no optimizer step has run yet) for
    # the HIP path and for the bf16-storage control alike, and hold the POOLED HIP error to the derived 1.5 x of the POOLED control error.
    NB = 8
    pool_hip = {k: [(grads[k].cpu().reshape(g_ref[k].shape), g_ref[k])] for k in ROUTER_PATH}
    pool_ctl = {k: [(g_ctl[k], g_ref[k])] for k in ROUTER_PATH}
    for sb in range(1, NB):
        gb = torch.Generator().manual_seed(7700 + sb)
        lat_b, img_b, noise_b = torch.randn(B, 4, 8, 8, generator=gb), torch.randn(B, 4, 8, 8, generator=gb) * 0.5, torch.randn(B, 4, 8, 8, generator=gb)
        t_b = torch.randint(0, 1000, (B,), generator=gb)
        ehs_b, ref_b = torch.randn(B, 5, 16, generator=gb), torch.randn(B, 9, 32, generator=gb)
        code_b = torch.randint(0, 5, (B,), generator=gb)
        batch_b = (lat_b, img_b, noise_b, t_b, ehs_b, ref_b, code_b, sa, s1)
        _, gr_b = oracle_training_grads(sd, cfg, prefixes, batch_b, control=False)
        _, gc_b = oracle_training_grads(sd, cfg, prefixes, batch_b, control=True)
        tr.zero_grad()   # `backward` ACCUMULATES across calls until an optimizer step (micro-batches): every batch of the pool starts a fresh accumulation
        _, tape_b, leaves_b = tr.forward_loss(lat_b.to(DEV), img_b.to(DEV), ehs_b.to(DEV), ref_b.to(DEV), code_b.to(DEV), noise_b.to(DEV), t_b.to(DEV))
        gh_b = tr.backward(tape_b, leaves_b)
        for k in ROUTER_PATH:
            pool_hip[k].append((gh_b[k].cpu().reshape(gr_b[k].shape), gr_b[k]))
            pool_ctl[k].append((gc_b[k], gr_b[k]))
    tr.zero_grad()       # (the step below is taken with the first batch's gradients, passed explicitly)
    for k in ROUTER_PATH:
        e_p, c_p = pooled_rel_l2(pool_hip[k]), pooled_rel_l2(pool_ctl[k])
        print(f"  grad {k:34s} pooled over {NB} batches: HIP {e_p:.3e}  control {c_p:.3e}  ratio {e_p / max(c_p, 1e-12):.2f}")
        assert e_p <= 1.5 * c_p + 1e-3, f"grad {k}: pooled HIP rel_l2 {e_p:.3e} vs pooled bf16-storage control {c_p:.3e}"
    # experts nobody was routed to get exactly zero gradient; routed ones do not
    _, top1, _ = A.task_gate(sd["task_embs"], code, sd["gate.weight"], sd["gate.bias"])
    g0 = grads["adapter_modules.0"].cpu()
    for e in range(g0.shape[0]):
        assert (float(g0[e].abs().max()) > 0) == (e in set(top1.tolist()))
    # one AdamW step moves the masters and lowers the loss on the same batch
    before = {k: moe.state_dict()[k].detach().clone() for k in names}
    tr.optimizer_step(grads)
    assert all(not torch.equal(before[k], moe.state_dict()[k]) for k in names if float(grads[k].abs().max()) > 0)
    for _ in range(5):
        l2 = tr.train_step(lat.to(DEV), img.to(DEV), ehs.to(DEV), ref_emb.to(DEV), code.to(DEV), noise.to(DEV), t.to(DEV))
    assert float(l2) < float(loss)
    # DDP-shaped path on one rank: gradients produced inside the exchange buckets are bit-identical to the plain path's, and the
    # buckets of the later layers were released while the backward of the earlier ones was still running
    moe2, _ = _tiny_moe()
    moe2 = moe2.to(DEV)
    trp = AnySDTrainer(moe2, sa.to(DEV), s1.to(DEV), lr=1e-3)
    moe3, _ = _tiny_moe()
    moe3 = moe3.to(DEV)
    trx = AnySDTrainer(moe3, sa.to(DEV), s1.to(DEV), lr=1e-3, always_exchange=True, bucket_bytes=1 << 12)
    args = (lat.to(DEV), img.to(DEV), ehs.to(DEV), ref_emb.to(DEV), code.to(DEV), noise.to(DEV), t.to(DEV))
    _, tp, lp = trp.forward_loss(*args)
    n_plain_nodes = len(tp.nodes)
    gp = trp.backward(tp, lp)
    _, tx, lx = trx.forward_loss(*args)
    gx = trx.backward(tx, lx)
    log = list(trx.exchange.launch_log)
    gx = trx.optimizer_step(gx)
    for k in gp:
        assert torch.equal(gp[k], gx[k].reshape(gp[k].shape)), k
    first_rs = next(i for i, e in enumerate(log) if e[0] == "reduce_scatter")
    assert first_rs < max(i for i, e in enumerate(log) if e[0] == "ready") and len(trx.exchange.buckets) > 2
    # gradient accumulation (train.py --gradient_accumulation_steps): two micro-batches before ONE optimizer step sum their
    # gradients — identically in the plain path and in the bucket path, where the first micro-batch runs under no-sync (nothing is sent)
    # and the buckets leave during the LAST backward; the optimizer then steps on the sum, not on the last micro-batch
    def fresh(**kw):
        m, _ = _tiny_moe()
        return AnySDTrainer(m.to(DEV), sa.to(DEV), s1.to(DEV), lr=1e-3, **kw)

    args_b = (lat.flip(0).to(DEV), img.to(DEV), ehs.flip(0).to(DEV), ref_emb.to(DEV), code.to(DEV), noise.flip(0).to(DEV), t.to(DEV))
    trq = fresh()
    _, tb, lb = trq.forward_loss(*args_b)
    gb_alone = {k: v.clone() for k, v in trq.backward(tb, lb).items()}
    for trainer in (fresh(), fresh(always_exchange=True, bucket_bytes=1 << 12)):
        _, t1, l1 = trainer.forward_loss(*args)
        trainer.backward(t1, l1, sync=False)
        if trainer.exchange is not None:
            assert not [e for e in trainer.exchange.launch_log if e[0] == "reduce_scatter"], "no-sync micro-batch: nothing may be sent"
        _, t2, l2 = trainer.forward_loss(*args_b)
        acc = trainer.backward(t2, l2, sync=True)
        if trainer.exchange is not None:
            sent = [e for e in trainer.exchange.launch_log if e[0] == "reduce_scatter"]
            assert len(sent) == len(trainer.exchange.buckets), "every bucket leaves exactly once, during the last micro-batch"
            acc = trainer.reduce_gradients()
        for k in gp:
            assert torch.equal(acc[k].reshape(gp[k].shape), gp[k] + gb_alone[k]), f"accumulated gradient of {k} is not the sum of the micro-batches"
        before = {k: trainer.params[k].detach().clone() for k in ("task_embs", "adapter_modules.0")}
        v0 = trainer.params["task_embs"]._version
        trainer.optimizer_step(acc)
        assert trainer._micro == 0 and all(not torch.equal(before[k], trainer.params[k].detach()) for k in before)
        assert trainer.params["task_embs"]._version > v0, "optimizer steps must bump the parameter version (packed-weight caches key on it)"
    # skipped step (overflow / NaN / gradient inspection): backward, NO optimizer step, zero_grad -> the next backward starts from zero
    # (without zero_grad the second backward would add to the first one's sums — that accumulation is the documented contract)
    trs = fresh()
    _, ts1, ls1 = trs.forward_loss(*args)
    g_first = {k: v.clone() for k, v in trs.backward(ts1, ls1).items()}
    trs.zero_grad()
    _, ts2, ls2 = trs.forward_loss(*args)
    g_again = trs.backward(ts2, ls2)
    for k in g_first:
        assert torch.equal(g_again[k], g_first[k]), f"{k}: a skipped step followed by zero_grad must not leave stale sums"
    _, ts3, ls3 = trs.forward_loss(*args)
    g_twice = trs.backward(ts3, ls3)          # no zero_grad in between: accumulates (micro-batches)
    k0 = "task_embs"
    assert torch.equal(g_twice[k0], g_first[k0] + g_first[k0])
    try4 = fresh(always_exchange=True, bucket_bytes=1 << 12)
    _, t3, l3 = try4.forward_loss(*args)
    g3 = try4.backward(t3, l3)
    with pytest.raises(ValueError):          # DDP: a dict of other tensors would be un-averaged local values
        try4.optimizer_step({k: v.clone() for k, v in g3.items()})
    try4.zero_grad()
    # backward/communication overlap: with the expert projections recorded next to their blocks, the first bucket leaves while tape
    # nodes of EARLIER blocks are still to run — i.e. before the backward pass has reached every adapter layer
    _, t4, l4 = try4.forward_loss(*args)
    replayed, at_launch = [], []
    t4.nodes = [(lambda f=f, i=i: (replayed.append(i), f())[1]) for i, f in enumerate(t4.nodes)]   # count tape nodes as they replay
    n_nodes = len(t4.nodes)
    launch = try4.exchange._launch
    try4.exchange._launch = lambda bi: (at_launch.append(len(replayed)), launch(bi))[1]
    try4.backward(t4, l4)
    try4.exchange._launch = launch
    assert len(at_launch) == len(try4.exchange.buckets) and len(replayed) == n_nodes
    assert at_launch[0] < n_nodes // 3, f"first reduce-scatter after {at_launch[0]} of {n_nodes} tape nodes: it must leave early in the backward pass"
    assert sum(1 for a in at_launch if a < n_nodes - 8) >= len(try4.moe.adapter_modules) - 1, "adapter buckets must leave while the UNet backward is running"
    try4.zero_grad()
    # the same step with the collectives really issued through RCCL ("nccl" backend, ONE rank: reduce-scatter / all-gather are identities):
    # drives the call sequence, the side stream, the async handles and the 256-byte-aligned bucket slots on the GPU stack the 8-GPU job uses
    import torch.distributed as dist
    if not dist.is_initialized():
        import os
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            moe5, _ = _tiny_moe()
            moe5 = moe5.to(DEV)
            trr = AnySDTrainer(moe5, sa.to(DEV), s1.to(DEV), lr=1e-3, always_exchange=True, bucket_bytes=1 << 12, force_collectives=True)
            assert trr.exchange.collect and trr.exchange.world == 1
            for _ in range(2):                                  # twice: persistent buckets and handles are reused across steps
                _, tr_, lr_ = trr.forward_loss(*args)
                trr.backward(tr_, lr_)
                gr = {k: v.clone() for k, v in trr.reduce_gradients().items()}
                trr.zero_grad()
                for k in gp:
                    assert torch.equal(gp[k], gr[k].reshape(gp[k].shape)), f"RCCL single-rank exchange changed {k}"
            torch.cuda.synchronize()
        finally:
            dist.destroy_process_group()
    # activation checkpointing (openaimodel.py:250, attention.py:268; util.py:102-143): every ResBlock / BasicTransformerBlock body is
    # dropped after the forward and recomputed in the backward — same kernels, so the loss is bit-identical; the gradients agree to
    # bf16 rounding only, because a segment sums its contributions to an outside tensor (skip / residual inputs) before handing
    # them over, a different order of bf16 additions from the flat tape's
    moe4, _ = _tiny_moe()
    n_ck = 0
    for m in moe4.modules():
        if hasattr(m, "use_checkpoint"):
            m.use_checkpoint = True
            n_ck += 1
        if m.__class__.__name__ == "BasicTransformerBlock":
            m.checkpoint = True
            n_ck += 1
    assert n_ck >= 6
    moe4 = moe4.to(DEV)
    trc = AnySDTrainer(moe4, sa.to(DEV), s1.to(DEV), lr=1e-3)
    lc, tc, lvc = trc.forward_loss(*args)
    lp2, _, _ = trp.forward_loss(*args)
    assert float(lc) == float(lp2)
    assert len(tc.nodes) < n_plain_nodes            # the block bodies are single nodes now
    gc = trc.backward(tc, lvc)
    for k in gp:
        # the router gradients are sums of small, cancelling per-layer terms (see the oracle comparison above), so rounding shows most there
        assert rel_l2(gc[k].reshape(gp[k].shape), gp[k]) <= (1e-1 if k.startswith("gate.") else 3e-2), f"checkpointed gradient differs: {k}"
