"""GPU parity tests of the composed path: transformer blocks, ResBlock, tiny UNet, DDIM sampler — HIP (bf16 activations,
fp32 accumulate) against the reference-derived fp32 goldens and the oracle.

Tolerance for composed bf16 graphs vs the fp32 reference: relative L2 <= 2e-2 and PSNR >= 38 dB (peak = dynamic range of
the reference output); DDIM integer bookkeeping bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, sub_sd, T, rel_l2, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, ref, rl2=2e-2, db=38.0, what=""):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    e, p = rel_l2(got, ref), psnr(got, ref)
    assert e <= rl2 and p >= db, f"{what}: rel_l2={e:.3e} (<= {rl2}), psnr={p:.1f} dB (>= {db})"


def test_transformer_blocks_golden():
    from anyedit_amd.ldm.modules.attention import BasicTransformerBlock, FeedForward, SpatialTransformer
    g = load_golden("transformer")
    blk = BasicTransformerBlock(64, 2, 32, context_dim=24, checkpoint=False)
    blk.load_state_dict(sub_sd(g, "btb.w."))
    close(blk.to(DEV)(T(g["btb.x"]).to(DEV), context=T(g["btb.ctx"]).to(DEV)), g["btb.y"], what="BasicTransformerBlock")
    ff = FeedForward(64, glu=True)
    ff.load_state_dict(sub_sd(g, "ff.w."))
    close(ff.to(DEV)(T(g["ff.x"]).to(DEV)), g["ff.y"], what="FeedForward/GEGLU")
    for tag, lin in (("st", False), ("st_lin", True)):
        st = SpatialTransformer(64, 2, 32, depth=1, context_dim=24, use_linear=lin, use_checkpoint=False)
        st.load_state_dict(sub_sd(g, f"{tag}.w."))
        close(st.to(DEV)(T(g[f"{tag}.x"]).to(DEV), context=T(g[f"{tag}.ctx"]).to(DEV)), g[f"{tag}.y"], what=f"SpatialTransformer {tag}")


def test_resblock_and_resampling_golden():
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample
    g = load_golden("resblock")
    for tag, (cin, cout) in {"same": (64, 64), "diff": (96, 64)}.items():
        rb = ResBlock(cin, 128, 0.0, out_channels=cout, dims=2)
        rb.load_state_dict(sub_sd(g, f"{tag}.w."))
        close(rb.to(DEV)(T(g[f"{tag}.x"]).to(DEV), T(g[f"{tag}.emb"]).to(DEV)), g[f"{tag}.y"], what=f"ResBlock {tag}")
    down = Downsample(64, True, dims=2, out_channels=64)
    down.load_state_dict(sub_sd(g, "down.w."))
    down = down.to(DEV)
    close(down(T(g["down.x"]).to(DEV)), g["down.y"], rl2=6e-3, what="Downsample")
    close(down(T(g["down.x7"]).to(DEV)), g["down.y7"], rl2=6e-3, what="Downsample odd size")
    up = Upsample(64, True, dims=2, out_channels=64)
    up.load_state_dict(sub_sd(g, "up.w."))
    close(up.to(DEV)(T(g["up.x"]).to(DEV)), g["up.y"], rl2=6e-3, what="Upsample")


def test_resample_and_scale_shift_kernels_vs_torch():
    """ae_resample2x_rows_bf16 (nearest x2: bit-exact copy; 2x2 mean: the fp32 mean of the four bf16 values rounded once, incl. odd sizes that floor) and
    ae_scale_shift_rows_bf16 (act(x (1 + scale) + shift), openaimodel.py:264-268) against the fp32 statements; refusals of malformed arguments."""
    from anyedit_amd import ops
    g = torch.Generator().manual_seed(7)
    for (B, H, W, C) in ((2, 6, 10, 32), (1, 7, 9, 64), (3, 16, 16, 320)):
        x = torch.randn(B, C, H, W, generator=g).bfloat16()
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
        up, Ho, Wo = ops.resample2x_rows(rows, B, H, W)
        ref = F.interpolate(x.float(), scale_factor=2, mode="nearest").bfloat16().permute(0, 2, 3, 1).reshape(-1, C)
        assert (Ho, Wo) == (2 * H, 2 * W) and torch.equal(up.cpu(), ref)
        dn, Ho, Wo = ops.resample2x_rows(rows, B, H, W, down=True)
        ref = F.avg_pool2d(x.float(), 2, 2).permute(0, 2, 3, 1).reshape(-1, C)
        assert (Ho, Wo) == (H // 2, W // 2) and dn.shape == ref.shape
        d = (dn.float().cpu() - ref).abs()
        assert float((d / (ref.abs() + 1e-3)).max()) <= 2.0 ** -8, "2x2 mean: within one bf16 rounding of the fp32 mean"
        HW = H * W
        emb = torch.randn(B, 2 * C + 8, generator=g).to(DEV)[:, 8:]        # strided rows: a column slice, as the batched projection hands it over
        for act in (True, False):
            y = ops.scale_shift_rows(rows, emb, B, HW, silu=act)
            sc, sh = emb[:, :C].cpu().repeat_interleave(HW, 0), emb[:, C:].cpu().repeat_interleave(HW, 0)
            ref = rows.float().cpu() * (1 + sc) + sh
            ref = F.silu(ref) if act else ref
            assert float(((y.float().cpu() - ref).abs() / (ref.abs() + 1e-2)).max()) <= 2.0 ** -7
    with pytest.raises(ValueError):
        ops.resample2x_rows(rows[:, :12].contiguous(), B, H, W)            # channels not a multiple of 8
    with pytest.raises(ValueError):
        ops.scale_shift_rows(rows, emb[:, :C], B, HW)                      # emb must hold scale | shift
    with pytest.raises(Exception):
        ops.resample2x_rows(torch.zeros(8, 8, dtype=torch.bfloat16, device=DEV), 8, 1, 1, down=True)   # a 2x2 mean of a 1x1 map


def test_unet_guided_diffusion_options_golden():
    """The constructor options VERDICT r5 listed as refused — ResBlock(up= / down= / use_scale_shift_norm=) (openaimodel.py:215-221, 254-268), Upsample /
    Downsample without a conv (:108-118, 152-155), UNetModel(resblock_updown=, use_scale_shift_norm=, conv_resample=False) (:600-616, 707-721), and the
    AttentionBlock UNet (use_spatial_transformer=False, :277-324, 344-409) — against the
    reference's outputs (tests/golden/unet_gd_tiny.npz, tools/gen_golden.py::gen_unet_gd); state-dict keys are the reference's."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import ResBlock, Downsample, Upsample, UNetModel
    from test_oracle_golden import GD_TINY
    g = load_golden("unet_gd_tiny")
    cases = {"up": (64, 32, True, False, False, False), "down": (32, 64, False, True, False, True), "ssn": (64, 64, False, False, True, False),
             "up_ssn": (32, 32, True, False, True, False), "down_ssn": (64, 96, False, True, True, False)}
    for tag, (cin, cout, up, down, ssn, use_conv) in cases.items():
        rb = ResBlock(cin, 128, 0.0, out_channels=cout, use_conv=use_conv, use_scale_shift_norm=ssn, dims=2, up=up, down=down)
        rb.load_state_dict(sub_sd(g, f"rb.{tag}.w."))          # strict: the key set is the reference's (h_upd / x_upd / AvgPool hold no parameters)
        close(rb.to(DEV)(T(g[f"rb.{tag}.x"]).to(DEV), T(g[f"rb.{tag}.emb"]).to(DEV)), g[f"rb.{tag}.y"], what=f"ResBlock {tag}")
    x = T(g["pool.x"]).to(DEV)
    close(Downsample(32, False, dims=2)(x), g["pool.down"], rl2=4e-3, what="Downsample(use_conv=False)")
    close(Downsample(32, False, dims=2)(T(g["pool.x7"]).to(DEV)), g["pool.down7"], rl2=4e-3, what="Downsample(use_conv=False), odd size")
    close(Upsample(32, False, dims=2)(x), g["pool.up"], rl2=4e-3, what="Upsample(use_conv=False)")
    x, t, ctx = T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["ctx"]).to(DEV)
    for tag, extra in {"updown_ssn": dict(resblock_updown=True, use_scale_shift_norm=True), "noconv": dict(conv_resample=False)}.items():
        with torch.device(DEV):
            unet = UNetModel(**dict(GD_TINY, **extra))
        unet.load_state_dict({k: v.to(DEV) for k, v in sub_sd(g, f"{tag}.w.").items()})
        unet.eval().requires_grad_(False)
        with torch.no_grad():
            y = unet(x, t, context=ctx)
            close(y, g[f"{tag}.y"], what=f"UNetModel {tag}")
            assert torch.equal(unet(x, t, context=ctx), y), "run-to-run bit-equal"
            emb = unet.time_embedding_rows(t)                   # the hoisted time-embedding pack carries 2 x Cout columns per scale-shift block
            y2 = unet.forward_rows(x, None, unet.context_rows(ctx), emb_pack=emb)
            assert torch.equal(y2, y), "hoisted time-embedding path = per-step path"
    # predict_codebook_ids (n_embed, openaimodel.py:731-736, 783-784): the "noconv" weights + the id_predictor head -> logits [B, 24, H, W]
    with torch.device(DEV):
        unet = UNetModel(**dict(GD_TINY, conv_resample=False, n_embed=24))
    unet.load_state_dict({k: v.to(DEV) for k, v in dict(sub_sd(g, "noconv.w."), **sub_sd(g, "codebook.w.")).items()})
    unet.eval().requires_grad_(False)
    with torch.no_grad():
        close(unet(x, t, context=ctx), g["codebook.y"], what="UNetModel n_embed (codebook logits)")
    # use_spatial_transformer=False: AttentionBlock layers (openaimodel.py:277-324; QKVAttentionLegacy / QKVAttention :344-409), no context
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import AttentionBlock
    from test_oracle_golden import GD_ADM
    for tag, (ch, kw) in {"legacy": (64, dict(num_heads=4)), "new": (64, dict(num_head_channels=16, use_new_attention_order=True)), "one_head": (32, dict())}.items():
        ab = AttentionBlock(ch, **kw)
        ab.load_state_dict(sub_sd(g, f"ab.{tag}.w."))
        close(ab.to(DEV)(T(g[f"ab.{tag}.x"]).to(DEV)), g[f"ab.{tag}.y"], what=f"AttentionBlock {tag}")
    for tag, extra in GD_ADM.items():
        with torch.device(DEV):
            unet = UNetModel(**dict(GD_TINY, **extra))
        unet.load_state_dict({k: v.to(DEV) for k, v in sub_sd(g, f"{tag}.w.").items()})
        unet.eval().requires_grad_(False)
        with torch.no_grad():
            y = unet(x, t)
            close(y, g[f"{tag}.y"], what=f"UNetModel {tag}")
            assert torch.equal(unet(x, t), y), "run-to-run bit-equal"


@pytest.mark.parametrize("tag,cin,cout,kw", [("up", 64, 64, dict(up=True)), ("up_ssn_skip", 64, 128, dict(up=True, use_scale_shift_norm=True)),
                                             ("down", 128, 128, dict(down=True)), ("down_ssn_conv_skip", 64, 128, dict(down=True, use_scale_shift_norm=True, use_conv=True)),
                                             ("ssn", 320, 320, dict(use_scale_shift_norm=True))])
def test_resblock_guided_diffusion_options_at_map_sizes_with_producer_statistics(tag, cin, cout, kw):
    """The same options as test_unet_guided_diffusion_options_golden at 32x32 / 64x64 maps, where the convs hand their output statistics to the next GroupNorm
    (HW > 256: the colstats epilogues, also behind the nearest-x2 gather with a time-embedding vector) — HIP against the oracle's restatement (pinned to the
    reference by the golden at tiny sizes) with its bf16-storage control: err(HIP) <= 1.5 x err(control)."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import ResBlock
    from oracle import ldm_ref as L
    torch.manual_seed(17)
    rb = ResBlock(cin, 256, 0.0, out_channels=cout, **kw)
    g = torch.Generator().manual_seed(18)
    with torch.no_grad():
        for p_ in rb.out_layers[-1].parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        for n_, p_ in rb.named_parameters():
            if "in_layers.0" in n_ or "out_layers.0" in n_:
                p_.copy_((1.0 if n_.endswith("weight") else 0.0) + 0.1 * torch.randn(p_.shape, generator=g))
            p_.copy_(p_.bfloat16().float())
    B, H, W = 2, 32, 32
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    emb = torch.randn(B, 256, generator=g)
    sd = {k: v.float() for k, v in rb.state_dict().items()}
    okw = dict(up=bool(kw.get("up")), down=bool(kw.get("down")), scale_shift=bool(kw.get("use_scale_shift_norm")))
    ref = L.resblock(sd, "", x, emb, **okw)
    with L.bf16_storage():
        ctl = L.resblock(sd, "", x, emb, **okw)
    with torch.no_grad():
        got = rb.to(DEV)(x.to(DEV), emb.to(DEV)).float().cpu()
    assert got.shape == ref.shape
    e, c = rel_l2(got, ref), rel_l2(ctl, ref)
    print(f"ResBlock {tag} @{H}x{W}: HIP rel-L2 {e:.3e}, bf16-storage control {c:.3e}")
    assert e <= 1.5 * c + 1e-4, (tag, e, c)


@pytest.mark.parametrize("tag,ch,hw,kw,heads,new", [("legacy_d64", 256, 32, dict(num_head_channels=64), 4, False), ("new_d32", 128, 24, dict(num_head_channels=32, use_new_attention_order=True), 4, True),
                                                     ("legacy_8_heads_d40", 320, 16, dict(num_heads=8), 8, False)])
def test_attention_block_at_map_sizes(tag, ch, hw, kw, heads, new):
    """AttentionBlock (openaimodel.py:277-324) at guided-diffusion sizes — up to 1 024 positions, head widths 32 / 40 / 64, both channel orders of the packed
    projection — HIP against the oracle restatement with its bf16-storage control."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import AttentionBlock
    from oracle import ldm_ref as L
    torch.manual_seed(19)
    ab = AttentionBlock(ch, **kw)
    assert ab.num_heads == heads
    g = torch.Generator().manual_seed(20)
    with torch.no_grad():
        for p_ in ab.proj_out.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
        ab.norm.weight.copy_(1.0 + 0.1 * torch.randn(ch, generator=g))
        ab.norm.bias.copy_(0.1 * torch.randn(ch, generator=g))
        ab.qkv.weight.mul_(3.0)                                  # logits of O(1): the softmax is not flat
        for p_ in ab.parameters():
            p_.copy_(p_.bfloat16().float())
    x = torch.randn(2, ch, hw, hw, generator=g).bfloat16().float()
    sd = {k: v.float() for k, v in ab.state_dict().items()}
    ref = L.attention_block(sd, "", x, heads, new_order=new)
    with L.bf16_storage():
        ctl = L.attention_block(sd, "", x, heads, new_order=new)
    with torch.no_grad():
        got = ab.to(DEV)(x.to(DEV)).float().cpu()
    h_ref = ref - x                                                # the block's own contribution (the residual passes through exactly)
    e, c = rel_l2(got - x, h_ref), rel_l2(ctl - x, h_ref)
    print(f"AttentionBlock {tag} ({hw * hw} positions): HIP rel-L2 of the attention branch {e:.3e}, bf16-storage control {c:.3e}")
    assert e <= 1.5 * c + 2e-3, (tag, e, c)


@pytest.mark.parametrize("cin,cout,hw", [(1280, 1280, 16), (2560, 1280, 8), (1280, 1280, 8)])
def test_resblock_small_maps_fold_splitk_into_groupnorm_bit_identical(cin, cout, hw):
    """ResBlock at the UNet's 16x16 / 8x8 levels (batch 12): conv1 -> (+ emb) -> GroupNorm + SiLU through the split-K fold (ops.conv3x3_partials + ops.groupnorm_splitk;
    opt-in, AE_GN_SPLITK=1: measured slower in the graph) equals the default reduce-launch path bit for bit, and launches one kernel fewer."""
    from anyedit_amd import ops
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import ResBlock
    torch.manual_seed(23)
    with torch.device(DEV):
        rb = ResBlock(cin, 1280, 0.0, out_channels=cout)
    g = torch.Generator().manual_seed(24)
    with torch.no_grad():
        for p_ in rb.out_layers[-1].parameters():
            p_.copy_((torch.randn(p_.shape, generator=g) * 0.02).to(DEV))
    B = 12
    x = torch.randn(B, cin, hw, hw, generator=g).to(DEV)
    emb = torch.randn(B, 1280, generator=g).to(DEV)
    knob = ops._GN_SPLITK
    try:
        with torch.no_grad():
            ops._GN_SPLITK = True
            assert ops.conv3x3_gn_splitk_ok(B, hw, hw, cin, cout)
            with ops.OpProfiler() as prof:
                y1 = rb(x, emb)
            n1 = sum(v["launches"] for v in prof.summary().values())
            assert any("splitK fold" in k for k in prof.summary())
            ops._GN_SPLITK = False
            with ops.OpProfiler() as prof:
                y0 = rb(x, emb)
            n0 = sum(v["launches"] for v in prof.summary().values())
            assert not any("splitK fold" in k for k in prof.summary())
    finally:
        ops._GN_SPLITK = knob
    assert torch.equal(y1, y0)
    assert n1 == n0 - 1, (n1, n0)     # (conv + reduce) + slab GroupNorm = 3 launches -> partial conv + folding GroupNorm = 2


@pytest.fixture(scope="module")
def tiny_unet():
    from util_models import build_tiny_unet
    g = load_golden("unet_tiny")
    unet = build_tiny_unet()
    unet.load_state_dict(sub_sd(g, "w."))
    return unet.to(DEV), g


def test_unet_tiny_golden(tiny_unet):
    unet, g = tiny_unet
    y = unet(T(g["x"]).to(DEV), T(g["t"]).to(DEV), context=T(g["ctx"]).to(DEV))
    close(y, g["y"], what="tiny UNet 8x8")
    y16 = unet(T(g["x16"]).to(DEV), T(g["t"])[:1].to(DEV), context=T(g["ctx"])[:1].to(DEV))
    close(y16, g["y16"], what="tiny UNet 16x16")


def test_unet_batch_independence_bit_exact(tiny_unet):
    """Sharding invariance (the multi-GPU data-parallel contract): a sample's output does not depend on what else is in
    the batch — bit-exact."""
    unet, g = tiny_unet
    x, t, ctx = T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["ctx"]).to(DEV)
    full = unet(x, t, context=ctx)
    for i in range(3):
        one = unet(x[i:i + 1], t[i:i + 1], context=ctx[i:i + 1])
        assert torch.equal(one[0], full[i]), f"sample {i} differs between batch-of-3 and batch-of-1"


def test_unet_kv_cache_and_determinism(tiny_unet):
    unet, g = tiny_unet
    x, t, ctx = T(g["x"]).to(DEV), T(g["t"]).to(DEV), T(g["ctx"]).to(DEV)
    rows = unet.context_rows(ctx)
    cache = {}
    a = unet.forward_rows(x, t, rows, kv_cache=cache)
    b = unet.forward_rows(x, t, rows, kv_cache=cache)  # second call hits the cached K|V projections
    c = unet.forward_rows(x, t, rows)
    assert len(cache) == 7 and torch.equal(a, b) and torch.equal(a, c)


def _tiny_ldm(unet):
    from anyedit_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    return LatentDiffusion(unet, conditioning_key="hybrid", timesteps=1000, linear_start=0.00085, linear_end=0.0120).to(DEV)


def test_apply_model_q_sample_mse(tiny_unet):
    unet, _ = tiny_unet
    g = load_golden("ddim_tiny")
    ldm = _tiny_ldm(unet)
    cond = {"c_concat": [T(g["img_lat"]).to(DEV)], "c_crossattn": [T(g["ctx"]).to(DEV)]}
    t = T(g["apply.t"]).to(DEV)
    close(ldm.apply_model(T(g["x_T"]).to(DEV), t, cond), g["apply.y"], what="apply_model hybrid")
    qs = ldm.q_sample(T(g["x_T"]).to(DEV), t, T(g["qs.noise"]).to(DEV))
    assert torch.equal(qs.cpu(), T(g["qs.y"])), "q_sample must be bit-exact (unfused fp32)"
    loss, _ = ldm.p_losses(T(g["x_T"]).to(DEV), cond, t, noise=T(g["qs.noise"]).to(DEV))
    assert abs(float(loss) - float(g["ploss.loss_simple"])) <= 2e-2 * float(g["ploss.loss_simple"])


def test_ddim_sampler_golden(tiny_unet):
    """Full sampler runs (CFG, mask/x0, eta=0) vs the reference's own sampler output; CPU noise stream replayed (G11)."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    unet, _ = tiny_unet
    g = load_golden("ddim_tiny")
    ldm = _tiny_ldm(unet)
    sampler = DDIMSampler(ldm)
    sampler.randn = lambda shape, device=None: torch.randn(shape).to(device)  # replay the CPU RNG stream of the golden run
    dev = lambda k: T(g[k]).to(DEV)
    cond = {"c_concat": [dev("img_lat")], "c_crossattn": [dev("ctx")]}
    uncond = {"c_concat": [dev("img_lat")], "c_crossattn": [dev("null_ctx")]}
    for tag, S, scale, use_mask, tol in (("s5_nocfg", 5, 1.0, False, 3e-2), ("s5_cfg", 5, 7.5, False, 6e-2),
                                         ("s20_cfg", 20, 7.5, False, 8e-2), ("s7_cfg_mask", 7, 3.0, True, 6e-2)):
        kw = dict(mask=dev(f"{tag}.mask"), x0=dev(f"{tag}.x0")) if use_mask else {}
        torch.manual_seed(1234)
        samples, inter = sampler.sample(S, 2, (4, 8, 8), cond, eta=0.0, x_T=dev("x_T"), verbose=False,
                                        unconditional_guidance_scale=scale,
                                        unconditional_conditioning=uncond if scale != 1.0 else None, log_every_t=1, **kw)
        assert np.array_equal(sampler.ddim_timesteps, g[f"{tag}.ddim_timesteps"])      # integer bookkeeping: bit-exact
        assert len(inter["x_inter"]) == g[f"{tag}.x_inter"].shape[0]
        close(samples, g[f"{tag}.samples"], rl2=tol, db=30.0, what=f"DDIM {tag}")       # absolute cap = north_star's "1e-3 PSNR-equivalent" (MSE / peak^2 <= 1e-3 <=> 30 dB); the derived bound below is what binds
        if not use_mask:
            # derived bound: the same sampler run of the oracle with bf16 STORAGE of activations / weights (fp32 arithmetic) is the
            # error any bf16-activation implementation must carry; the HIP path may add at most half of it again
            from oracle import ldm_ref as L, ddim_ref as D, schedule_ref as SR
            from util_models import TINY_UNET
            sdb = L.bf16_weights(sub_sd(load_golden("unet_tiny"), "w."))
            buffers = SR.register_schedule("linear", 1000, 0.00085, 0.0120)
            cpu = lambda k: T(g[k])
            c_cpu = {"c_concat": [cpu("img_lat")], "c_crossattn": [cpu("ctx")]}
            u_cpu = {"c_concat": [cpu("img_lat")], "c_crossattn": [cpu("null_ctx")]}
            with torch.no_grad(), L.bf16_storage():
                ctl, _, _ = D.ddim_sample(lambda x, t, c: L.diffusion_wrapper(sdb, TINY_UNET, x, t, c["c_concat"], c["c_crossattn"], "hybrid"),
                                          buffers, S, (2, 4, 8, 8), c_cpu, eta=0.0, x_T=cpu("x_T"), scale=scale,
                                          uc=u_cpu if scale != 1.0 else None)
            e_hip, e_ctl = rel_l2(samples.float().cpu(), T(g[f"{tag}.samples"])), rel_l2(ctl, T(g[f"{tag}.samples"]))
            assert e_hip <= 1.5 * e_ctl + 1e-3, f"DDIM {tag}: HIP {e_hip:.3e} vs bf16-storage control {e_ctl:.3e}"


def test_ddim_encode_inversion_golden(tiny_unet):
    """DDIMSampler.encode on the HIP path vs the reference's inversion output (golden) and its integer bookkeeping."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    unet, _ = tiny_unet
    g = load_golden("ddim_encode")
    ldm = _tiny_ldm(unet)
    sampler = DDIMSampler(ldm)
    sampler.make_schedule(10, ddim_eta=0.0, verbose=False)
    dev = lambda k: T(g[k]).to(DEV)
    cond = {"c_concat": [dev("img_lat")], "c_crossattn": [dev("ctx")]}
    x_enc, out = sampler.encode(dev("x0"), cond, t_enc=7, return_intermediates=3)
    assert out["intermediate_steps"] == g["intermediate_steps"].tolist()
    close(x_enc, g["x_encoded"], rl2=3e-2, db=30.0, what="DDIM inversion (7 of 10 steps)")
    close(torch.stack(out["intermediates"]), g["intermediates"], rl2=3e-2, db=30.0, what="DDIM inversion intermediates")
    x_enc2, _ = sampler.encode(dev("x0"), cond, t_enc=20, use_original_steps=True)
    close(x_enc2, g["x_encoded_original_steps"], rl2=3e-2, db=30.0, what="DDIM inversion (original steps)")


def test_ddim_encode_step_bit_exact_vs_oracle_arithmetic():
    """Same eps on both sides: the fused inversion update reproduces the reference's un-fused fp32 expression bit for bit,
    including the CFG combination."""
    from anyedit_amd import ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 4, 8, 8, generator=g)
    eps = torch.randn(4, 4, 8, 8, generator=g)
    a_next = torch.tensor(0.8531, dtype=torch.float32)
    a_prev = torch.tensor(0.9127, dtype=torch.float64)
    cx = (a_next / a_prev).sqrt()
    ce = a_next.sqrt() * ((1 / a_next - 1).sqrt() - (1 / a_prev - 1).sqrt())
    eu, ec = eps.chunk(2)
    e = eu + 7.5 * (ec - eu)
    ref = cx * x + ce * e
    got = ops.ddim_encode_step(x.to(DEV), eps.to(DEV), float(cx.float()), float(ce.float()), branches=2, scale=7.5)
    assert torch.equal(got.cpu(), ref.float())


def test_plms_sampler_golden():
    """PLMSSampler on the HIP path vs the reference sampler (golden from the analytic eps model: identical eps on both sides, so
    this isolates the sampler arithmetic, the eps history, the integer bookkeeping and the RNG consumption order)."""
    from anyedit_amd.ldm.models.diffusion.plms import PLMSSampler
    from oracle import schedule_ref as S

    class AnalyticModel:
        parameterization = "eps"

        def __init__(self, dev):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)

        def apply_model(self, x, t, c):
            xc, tc, cc = x.detach().float().cpu(), t.cpu(), c.float().cpu()
            return (torch.sin(xc * 1.7 + tc.float()[:, None, None, None] * 0.01) * 0.5 + cc[:, :, None, None] * xc).to(x.device)

    g = load_golden("plms")
    sampler = PLMSSampler(AnalyticModel(DEV))
    sampler.randn = lambda shape, device=None: torch.randn(shape).to(device)        # replay the CPU RNG stream of the golden run
    dev = lambda k: T(g[k]).to(DEV)
    for tag, steps, scale, use_mask in (("s7", 7, 1.0, False), ("s10_cfg", 10, 5.0, False), ("s6_cfg_mask", 6, 3.0, True)):
        kw = dict(mask=dev(f"{tag}.mask"), x0=dev(f"{tag}.x0")) if use_mask else {}
        torch.manual_seed(4321)
        samples, inter = sampler.sample(steps, 2, (4, 8, 8), dev("c"), eta=0.0, x_T=dev("x_T"), verbose=False,
                                        unconditional_guidance_scale=scale, unconditional_conditioning=dev("uc") if scale != 1.0 else None,
                                        log_every_t=1, **kw)
        assert np.array_equal(sampler.ddim_timesteps, g[f"{tag}.ddim_timesteps"])
        assert len(inter["x_inter"]) == g[f"{tag}.x_inter"].shape[0]
        assert float((samples.cpu() - T(g[f"{tag}.samples"])).abs().max()) <= 2e-5, tag
        assert float((torch.stack(inter["pred_x0"]).cpu() - T(g[f"{tag}.pred_x0"])).abs().max()) <= 2e-5, tag

    class Corrector:   # tools/gen_golden.py::AnalyticCorrector (plms.py:195-197); noise_dropout (:222-224) only moves the RNG at eta = 0
        def modify_score(self, model, e_t, x, t, c, gain=1.0):
            return e_t * gain - 0.05 * x + 0.01 * c[:, :, None, None]

    torch.manual_seed(4321)
    samples, inter = sampler.sample(6, 2, (4, 8, 8), dev("c"), eta=0.0, x_T=dev("x_T"), verbose=False, unconditional_guidance_scale=3.0,
                                    unconditional_conditioning=dev("uc"), log_every_t=1, score_corrector=Corrector(), corrector_kwargs={"gain": 1.1}, noise_dropout=0.3)
    assert float((samples.cpu() - T(g["s6_cfg_corr.samples"])).abs().max()) <= 2e-5
    assert float((torch.stack(inter["pred_x0"]).cpu() - T(g["s6_cfg_corr.pred_x0"])).abs().max()) <= 2e-5
    with pytest.raises(ValueError):
        sampler.make_schedule(5, ddim_eta=0.5, verbose=False)


def test_ddim_sampler_v_prediction_golden():
    """DDIMSampler on a v-prediction model (ddim.py:214-217, 232-235) and DDPM's predict_start_from_z_and_v / predict_eps_from_z_and_v /
    get_v (ddpm.py:290-302, 361-365) against the reference (golden from the analytic network read as v: identical network output on both
    sides isolates the sampler arithmetic).  fp32 kernels, 2e-5."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM
    from oracle import schedule_ref as S

    class AnalyticVModel:
        parameterization = "v"
        _acp_pair = DDPM._acp_pair
        predict_start_from_z_and_v = DDPM.predict_start_from_z_and_v
        predict_eps_from_z_and_v = DDPM.predict_eps_from_z_and_v
        get_v = DDPM.get_v

        def __init__(self, dev):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)

        def apply_model(self, x, t, c):
            xc, tc, cc = x.detach().float().cpu(), t.cpu(), c.float().cpu()
            return (torch.sin(xc * 1.7 + tc.float()[:, None, None, None] * 0.01) * 0.5 + cc[:, :, None, None] * xc).to(x.device)

    g = load_golden("ddim_v")
    model = AnalyticVModel(DEV)
    sampler = DDIMSampler(model)
    sampler.randn = lambda shape, device=None: torch.randn(shape).to(device)        # replay the CPU RNG stream of the golden run
    dev = lambda k: T(g[k]).to(DEV)
    for tag, steps, scale, eta in (("s6", 6, 1.0, 0.0), ("s8_cfg", 8, 5.0, 0.0), ("s5_cfg_eta1", 5, 3.0, 1.0)):
        torch.manual_seed(4323)
        samples, inter = sampler.sample(steps, 2, (4, 8, 8), dev("c"), eta=eta, x_T=dev("x_T"), verbose=False,
                                        unconditional_guidance_scale=scale, unconditional_conditioning=dev("uc") if scale != 1.0 else None,
                                        log_every_t=1)
        assert np.array_equal(sampler.ddim_timesteps, g[f"{tag}.ddim_timesteps"])
        assert float((samples.cpu() - T(g[f"{tag}.samples"])).abs().max()) <= 2e-5, tag
        assert float((torch.stack(inter["pred_x0"]).cpu() - T(g[f"{tag}.pred_x0"])).abs().max()) <= 2e-5, tag
    x, t, noise, v = dev("x_T"), dev("t"), dev("noise"), dev("v")
    assert float((model.get_v(x, noise, t).cpu() - T(g["get_v"])).abs().max()) <= 1e-6
    assert float((model.predict_start_from_z_and_v(x, t, v).cpu() - T(g["x0_from_v"])).abs().max()) <= 1e-6
    assert float((model.predict_eps_from_z_and_v(x, t, v).cpu() - T(g["eps_from_v"])).abs().max()) <= 1e-6
    model.parameterization = "x0"
    with pytest.raises(NotImplementedError):
        sampler.sample(2, 2, (4, 8, 8), dev("c"), x_T=dev("x_T"), verbose=False)


def test_ddim_hacked_sampler_golden():
    """cldm.ddim_hacked.DDIMSampler (AnyDoor path) vs the reference's: two network calls per guided step, inversion at ddim_timesteps[i]."""
    from anyedit_amd.cldm.ddim_hacked import DDIMSampler
    from oracle import schedule_ref as S

    class AnalyticModel:
        parameterization = "eps"

        def __init__(self, dev):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)
            self.calls, self.seen_t = 0, []

        def apply_model(self, x, t, c):
            self.calls += 1
            self.seen_t.append(int(t[0]))
            xc, tc, cc = x.detach().float().cpu(), t.float().cpu(), c.float().cpu()
            return (torch.sin(xc * 1.7 + tc[:, None, None, None] * 0.01) * 0.5 + cc[:, :, None, None] * xc).to(x.device)

    g = load_golden("ddim_hacked")
    model = AnalyticModel(DEV)
    sampler = DDIMSampler(model)
    sampler.randn = lambda shape, device=None: torch.randn(shape).to(device)
    dev = lambda k: T(g[k]).to(DEV)
    torch.manual_seed(4322)
    samples, inter = sampler.sample(8, 2, (4, 8, 8), dev("c"), eta=0.0, x_T=dev("x_T"), verbose=False, unconditional_guidance_scale=5.0,
                                    unconditional_conditioning=dev("uc"), log_every_t=1)
    assert model.calls == int(g["s8_cfg.network_calls"]) and np.array_equal(sampler.ddim_timesteps, g["ddim_timesteps"])
    assert float((samples.cpu() - T(g["s8_cfg.samples"])).abs().max()) <= 2e-5
    assert float((torch.stack(inter["pred_x0"]).cpu() - T(g["s8_cfg.pred_x0"])).abs().max()) <= 2e-5
    model.seen_t = []
    x, out = sampler.encode(dev("x_T"), dev("c"), t_enc=6, return_intermediates=2)
    assert model.seen_t == [int(v) for v in g["ddim_timesteps"][:6]]                    # queried at the schedule's timesteps
    assert out["intermediate_steps"] == g["enc.intermediate_steps"].tolist()
    assert float((x.cpu() - T(g["enc.x"])).abs().max()) <= 2e-5
    x, _ = sampler.encode(dev("x_T"), dev("c"), t_enc=6, unconditional_guidance_scale=3.0, unconditional_conditioning=dev("uc"))
    assert float((x.cpu() - T(g["enc.x_cfg"])).abs().max()) <= 2e-5
    x, _ = sampler.encode(dev("x_T"), dev("c"), t_enc=15, use_original_steps=True)
    assert float((x.cpu() - T(g["enc.x_orig"])).abs().max()) <= 2e-5


def test_unet_class_conditional_and_adm_keys_golden():
    """Class-conditional UNet (label_emb, openaimodel.py:533-535, 764-772) behind DiffusionWrapper's 'hybrid-adm' / 'crossattn-adm' keys
    (ddpm.py:1349-1358) against the reference's outputs; bf16-storage tolerance of the tiny UNet tests."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from anyedit_amd.ldm.models.diffusion.ddpm import DiffusionWrapper
    from test_oracle_golden import SD2_TINY
    g = load_golden("unet_adm_tiny")
    with torch.device(DEV):
        unet = UNetModel(**dict(SD2_TINY, num_classes=5, in_channels=6))
    unet.load_state_dict({k: v.to(DEV) for k, v in sub_sd(g, "w.").items()})
    unet.eval().requires_grad_(False)
    dev = lambda k: T(g[k]).to(DEV)
    x, cc, t, ctx, y = dev("x"), dev("cc"), dev("t"), dev("ctx"), dev("y")
    w = DiffusionWrapper(unet, "hybrid-adm")
    with torch.no_grad():
        out = w(x, t, c_concat=[cc], c_crossattn=[ctx], c_adm=y)
        assert rel_l2(out.cpu(), T(g["out.hybrid_adm"])) <= 2e-2
        w.conditioning_key = "crossattn-adm"
        out2 = w(torch.cat([x, cc], 1), t, c_crossattn=[ctx[:, :4], ctx[:, 4:]], c_adm=y)
        assert rel_l2(out2.cpu(), T(g["out.crossattn_adm"])) <= 2e-2
        other = w(torch.cat([x, cc], 1), t, c_crossattn=[ctx], c_adm=(y + 1) % 5)
        assert rel_l2(other.cpu(), out2.cpu()) > 1e-3                      # the label matters
        with pytest.raises(AssertionError):
            unet(torch.cat([x, cc], 1), t, context=ctx)                  # class-conditional model without y
    # num_classes = "continuous" (openaimodel.py:536-538): Linear(1, 4*mc) over a real-valued y [B, 1], same network otherwise
    with torch.device(DEV):
        unet_c = UNetModel(**dict(SD2_TINY, num_classes="continuous", in_channels=6))
    unet_c.load_state_dict({k: v.to(DEV) for k, v in dict(sub_sd(g, "w."), **sub_sd(g, "wc.")).items()})
    unet_c.eval().requires_grad_(False)
    with torch.no_grad():
        out3 = DiffusionWrapper(unet_c, "crossattn-adm")(torch.cat([x, cc], 1), t, c_crossattn=[ctx], c_adm=dev("y_cont"))
        lab = unet_c._label_rows(dev("y_cont"), DEV)
    assert rel_l2(out3.cpu(), T(g["out.continuous"])) <= 2e-2
    ref_lab = torch.nn.functional.linear(T(g["y_cont"]), sub_sd(g, "wc.")["label_emb.weight"].float(), sub_sd(g, "wc.")["label_emb.bias"].float())
    assert rel_l2(lab.cpu(), ref_lab) <= 1e-6                            # the label embedding itself: exact-fp32 GEMM
    with pytest.raises(ValueError):
        UNetModel(**dict(SD2_TINY, num_classes="sequential", in_channels=6))


def test_dpm_solver_sampler_golden():
    """DPMSolverSampler (DPM-Solver++ 2M, CFG) and the other multistep variants on the HIP path vs the reference solver (golden from the
    analytic eps model: identical eps on both sides isolates the solver arithmetic, history handling and step bookkeeping)."""
    from anyedit_amd.ldm.models.diffusion.dpm_solver import DPMSolverSampler
    from anyedit_amd.ldm.models.diffusion.dpm_solver.dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver
    from oracle import schedule_ref as S

    class AnalyticModel:
        parameterization = "eps"

        def __init__(self, dev):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)
            self.calls = 0

        def apply_model(self, x, t, c):
            self.calls += 1
            xc, tc, cc = x.detach().float().cpu(), t.float().cpu(), c.float().cpu()
            return (torch.sin(xc * 1.7 + tc[:, None, None, None] * 0.01) * 0.5 + cc[:, :, None, None] * xc).to(x.device)

    def near(got, key, rel=5e-6):                                            # fp32 round-off relative to the sample magnitude
        ref = T(g[key])
        assert float((got.cpu() - ref).abs().max()) <= rel * max(float(ref.abs().max()), 1.0), key

    g = load_golden("dpm_solver")
    model = AnalyticModel(DEV)
    sampler = DPMSolverSampler(model)
    dev = lambda k: T(g[k]).to(DEV)
    for tag, steps, scale in (("s10", 10, 1.0), ("s12_cfg", 12, 5.0), ("s20_cfg", 20, 7.5)):
        model.calls = 0
        samples, none = sampler.sample(steps, 2, (4, 8, 8), dev("c"), x_T=dev("x_T"), verbose=False, unconditional_guidance_scale=scale,
                                       unconditional_conditioning=dev("uc") if scale != 1.0 else None)
        assert none is None and model.calls == steps                      # one network evaluation per step
        near(samples, f"{tag}.samples")
    ns = NoiseScheduleVP('discrete', alphas_cumprod=model.alphas_cumprod)
    tq = T(g["ns.t"])
    assert float((ns.marginal_lambda(tq) - T(g["ns.lambda"])).abs().max()) <= 1e-5
    assert float((ns.inverse_lambda(ns.marginal_lambda(tq)) - T(g["ns.inverse_lambda"])).abs().max()) <= 1e-6
    mf = model_wrapper(lambda x, t, c: model.apply_model(x, t, c), ns, model_type="noise", guidance_type="classifier-free",
                       condition=dev("c"), unconditional_condition=dev("uc"), guidance_scale=3.0)
    for st in ("time_uniform", "logSNR", "time_quadratic"):
        assert float((DPM_Solver(mf, ns).get_time_steps(st, 1.0, 0.001, 10, DEV) - T(g[f"ts.{st}"])).abs().max()) <= 1e-6
    out = DPM_Solver(mf, ns, predict_x0=False).sample(dev("x_T"), steps=9, skip_type="logSNR", method="multistep", order=2)
    near(out, "eps2m.samples")
    out = DPM_Solver(mf, ns, predict_x0=True).sample(dev("x_T"), steps=8, skip_type="time_quadratic", method="multistep", order=2,
                                                     solver_type="taylor", denoise_to_zero=True)
    near(out, "taylor.samples")
    out = DPM_Solver(mf, ns, predict_x0=True).sample(dev("x_T"), steps=6, skip_type="time_uniform", method="multistep", order=1,
                                                     t_start=0.8, t_end=0.05)
    near(out, "o1.samples")
    # the wrapped model is still callable the reference's way: guided noise at a continuous time
    x, t = dev("x_T"), torch.tensor([0.5])
    eu = model.apply_model(x, torch.full((2,), 499.0), dev("uc"))
    ec = model.apply_model(x, torch.full((2,), 499.0), dev("c"))
    assert float((mf(x, t) - (eu + 3.0 * (ec - eu))).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        DPM_Solver(mf, ns).sample(x, steps=6, method="no_such_method")
    with pytest.raises(NotImplementedError):
        NoiseScheduleVP('linear')


def test_dpm_solver_general_variants_golden():
    """DPM_Solver.sample beyond the sampler front end's settings — singlestep orders 2 / 3 with the order plan, singlestep_fixed, multistep
    order 3, adaptive step size with its evaluation count — on the HIP path (network evaluation through ae_dpm_multistep_f32, every update one
    ae_lincomb4_f32 launch, the adaptive error norm ae_dpm_adaptive_err_f32) against outputs of the reference solver on the analytic eps
    model.  Samples grow to |x| ~ 5e2 under that toy network: the bound is relative; the update coefficients are applied to the model
    values themselves rather than to their differences, which costs a few fp32 ulps of |m|."""
    from anyedit_amd.ldm.models.diffusion.dpm_solver.dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver
    from oracle import schedule_ref as S
    from dpm_cases import DPM_GENERAL_CASES
    g = load_golden("dpm_solver_general")
    ac = S.register_schedule("linear", 1000, 0.00085, 0.0120)["alphas_cumprod"].float()

    def apply_model(x, t, c):
        xc, tc, cc = x.detach().float().cpu(), t.float().cpu(), c.float().cpu()
        return (torch.sin(xc * 1.7 + tc[:, None, None, None] * 0.01) * 0.5 + cc[:, :, None, None] * xc).to(x.device)

    ns = NoiseScheduleVP('discrete', alphas_cumprod=ac)
    dev = lambda k: T(g[k]).to(DEV)
    mf = model_wrapper(apply_model, ns, model_type="noise", guidance_type="classifier-free", condition=dev("c"),
                       unconditional_condition=dev("uc"), guidance_scale=3.0)
    plan = DPM_Solver(mf, ns)
    for steps, order in ((10, 3), (9, 3), (11, 3), (7, 2), (6, 2), (5, 1)):
        ts, orders = plan.get_orders_and_timesteps_for_singlestep_solver(steps, order, "logSNR", 1.0, 0.001, DEV)
        assert list(orders) == list(g[f"plan.{steps}.{order}.logSNR.orders"])
        assert float((ts.cpu() - T(g[f"plan.{steps}.{order}.logSNR.ts"])).abs().max()) <= 1e-6
    for tag, (px0, kw) in DPM_GENERAL_CASES.items():
        solver = DPM_Solver(mf, ns, predict_x0=px0)
        out = solver.sample(dev("x_T"), **kw)
        ref = T(g[f"{tag}.samples"])
        err = float((out.cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-5, (tag, err)
        if kw["method"] == "adaptive":
            assert solver.last_nfe == int(g[f"{tag}.nfe"]), (tag, solver.last_nfe)
    # paths the reference cannot run (it raises): its evident intent
    out = DPM_Solver(mf, ns, predict_x0=True).sample(dev("x_T"), steps=8, order=3, method="multistep")          # lower_order_final, < 15 steps
    assert torch.isfinite(out).all()
    out = DPM_Solver(mf, ns).sample(dev("x_T"), steps=10, order=3, method="singlestep", skip_type="time_uniform")
    assert torch.isfinite(out).all()
    out1 = DPM_Solver(mf, ns, predict_x0=True).sample(dev("x_T"), steps=5, order=1, method="singlestep", skip_type="time_uniform")
    outm = DPM_Solver(mf, ns, predict_x0=True).sample(dev("x_T"), steps=5, order=1, method="multistep", skip_type="time_uniform")
    assert float((out1 - outm).abs().max()) <= 2e-5 * float(outm.abs().max())     # order 1 is DDIM either way


def test_dpm_solver_tiny_unet_vs_oracle(tiny_unet):
    """DPMSolverSampler over the HIP UNet (hybrid dict conditioning, CFG 5, float timesteps) vs the oracle solver over the oracle UNet."""
    from anyedit_amd.ldm.models.diffusion.dpm_solver import DPMSolverSampler
    from oracle import dpm_ref as P, ldm_ref as L, schedule_ref as S
    from util_models import TINY_UNET
    unet, gu = tiny_unet
    g = load_golden("ddim_tiny")
    ldm = _tiny_ldm(unet)
    dev = lambda k: T(g[k]).to(DEV)
    cond = {"c_concat": [dev("img_lat")], "c_crossattn": [dev("ctx")]}
    uncond = {"c_concat": [dev("img_lat")], "c_crossattn": [dev("null_ctx")]}
    samples, _ = DPMSolverSampler(ldm).sample(12, 2, (4, 8, 8), cond, x_T=dev("x_T"), verbose=False, unconditional_guidance_scale=5.0,
                                              unconditional_conditioning=uncond)
    sd = sub_sd(gu, "w.")
    apply_model = lambda x, t, c: L.diffusion_wrapper(sd, TINY_UNET, x, t, c["c_concat"], c["c_crossattn"], "hybrid")
    cpu = lambda c: {k: [v.cpu() for v in vs] for k, vs in c.items()}
    ac = S.register_schedule("linear", 1000, 0.00085, 0.0120)["alphas_cumprod"].float()
    ref = P.multistep_sample(apply_model, ac, T(g["x_T"]), 12, cpu(cond), cpu(uncond), 5.0)
    close(samples, ref, rl2=6e-2, db=30.0, what="DPM-Solver++ 2M, 12 steps, tiny UNet")


def test_unet_sd2_options_golden():
    """UNetModel with num_head_channels + use_linear_in_transformer (AnyDoor / SD-2.1 options) vs the reference's output."""
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    g = load_golden("unet_sd2_tiny")
    unet = UNetModel(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[1, 2],
                     channel_mult=[1, 2], num_head_channels=16, use_spatial_transformer=True, use_linear_in_transformer=True,
                     transformer_depth=1, context_dim=24, legacy=False)
    sd = sub_sd(g, "w.")
    assert set(unet.state_dict().keys()) == set(sd.keys())
    unet.load_state_dict(sd)
    y = unet.to(DEV)(T(g["x"]).to(DEV), T(g["t"]).to(DEV), context=T(g["ctx"]).to(DEV))
    close(y, g["y"], what="tiny UNet, head width 16, linear projections")


def test_anydoor_geometry_runs():
    """ControlledUnetModel + ControlNet at the AnyDoor sizes (anydoor.yaml:22-54: 320-wide, heads of 64, Linear transformer
    projections, 1024-wide context of 257 DINOv2 tokens) at 32x32 latents: shapes, finiteness, 13 control residuals."""
    from anyedit_amd.cldm.cldm import ControlNet, ControlledUnetModel
    geo = dict(image_size=32, in_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
               transformer_depth=1, context_dim=1024, legacy=False)
    torch.manual_seed(0)
    with torch.device(DEV):
        unet = ControlledUnetModel(out_channels=4, **geo)
        cnet = ControlNet(hint_channels=4, **geo)
    with torch.no_grad():
        for m in (unet, cnet):
            for p in m.parameters():
                if float(p.abs().max()) == 0.0:
                    p.normal_(0, 0.02)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 32, 32, generator=gen).to(DEV)
    hint = torch.rand(2, 4, 256, 256, generator=gen).to(DEV)
    ctx = torch.randn(2, 257, 1024, generator=gen).to(DEV)
    t = torch.tensor([981, 21], device=DEV)
    with torch.no_grad():
        control = cnet(x=x, hint=hint, timesteps=t, context=ctx)
        eps = unet(x=x, timesteps=t, context=ctx, control=control, only_mid_control=False)
        plain = unet(x=x, timesteps=t, context=ctx, control=None)
    assert len(control) == 13 and control[0].shape == (2, 320, 32, 32) and control[-1].shape == (2, 1280, 4, 4)
    assert eps.shape == (2, 4, 32, 32) and torch.isfinite(eps).all() and torch.isfinite(plain).all()
    assert float((eps - plain).abs().max()) > 0.0
    with torch.no_grad():
        again = unet(x=x[:1], timesteps=t[:1], context=ctx[:1], control=[c[:1] for c in control], only_mid_control=False)
    close(again, eps[:1].float().cpu(), rl2=1e-2, db=40.0, what="batch independence")   # split-K plans differ with M: bf16 round-off only


def test_ddim_sampler_vs_oracle_same_eps():
    """With the SAME eps fed to both, the HIP sampler arithmetic is bit-identical to the oracle's fp32 loop."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from oracle import ddim_ref as D, schedule_ref as S

    class FakeModel:
        parameterization = "eps"

        def __init__(self, dev):
            b = S.register_schedule("linear", 1000, 0.00085, 0.0120)
            self.num_timesteps = 1000
            for k, v in b.items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)

        def apply_model(self, x, t, c):
            # deterministic "network": a fixed elementwise function of (x, t, c) computed in fp32 on the CPU, so both sides see
            # bit-identical eps
            xc = x.detach().float().cpu()
            out = torch.sin(xc * 1.7 + t.cpu().float()[:, None, None, None] * 0.01) * 0.5 + c.cpu().float()[:, :, None, None] * xc
            return out.to(x.device)

    B = 2
    x_T = torch.randn(B, 4, 8, 8, generator=torch.Generator().manual_seed(3))
    c, uc = torch.full((B, 1), 0.3), torch.full((B, 1), -0.2)
    m_cpu = FakeModel("cpu")
    buffers = S.register_schedule("linear", 1000, 0.00085, 0.0120)
    torch.manual_seed(99)
    ref, _, _ = D.ddim_sample(m_cpu.apply_model, buffers, 50, tuple(x_T.shape), c, eta=0.3, x_T=x_T, scale=5.0, uc=uc)
    m = FakeModel(DEV)
    s = DDIMSampler(m)
    s.randn = lambda shape, device=None: torch.randn(shape).to(device)
    torch.manual_seed(99)
    got, _ = s.sample(50, B, (4, 8, 8), c.to(DEV), eta=0.3, x_T=x_T.to(DEV), verbose=False, unconditional_guidance_scale=5.0,
                      unconditional_conditioning=uc.to(DEV))
    # torch-CPU sqrt of the schedule scalars may differ from the correctly rounded one by 1 ulp (see test_host_logic), which
    # perturbs the trajectory at the 1e-7 level; everything else is bit-identical arithmetic
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_p_sample_ddim_score_corrector_and_noise_dropout():
    """ddim.py:219-221, 246-247: the two p_sample_ddim options this mirror refused until round 6.  score_corrector.modify_score sees the guidance-combined eps and its
    return value drives the update; noise_dropout drops (and rescales) the step noise.  Against the unfused fp32 statements of the reference on the same eps / noise / mask."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from oracle import schedule_ref as S

    class FakeModel:
        parameterization = "eps"

        def __init__(self, dev):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(dev))
            self.device = torch.device(dev)

        def apply_model(self, x, t, c):
            return torch.sin(x.float() * 1.7) * 0.5 + c.float()[:, :, None, None] * x.float()

    class Corrector:
        def modify_score(self, model, e_t, x, t, c, gain=1.0):
            self.seen = e_t.clone()
            return e_t * gain - 0.1 * x

    B = 2
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 4, 8, 8, generator=g)
    c, uc = torch.full((B, 1), 0.3), torch.full((B, 1), -0.2)
    noise = torch.randn(B, 4, 8, 8, generator=g)
    s = DDIMSampler(FakeModel(DEV))
    s.make_schedule(ddim_num_steps=20, ddim_eta=0.5, verbose=False)
    s.randn = lambda shape, device=None: noise.to(device)
    index, scale, temp, p_drop = 7, 4.0, 0.9, 0.25
    t = torch.full((B,), int(s.ddim_timesteps[index]), dtype=torch.long, device=DEV)
    cor = Corrector()
    torch.manual_seed(5)
    x_prev, pred_x0 = s.p_sample_ddim(x.to(DEV), c.to(DEV), t, index, temperature=temp, noise_dropout=p_drop, score_corrector=cor, corrector_kwargs={"gain": 1.2},
                                      unconditional_guidance_scale=scale, unconditional_conditioning=uc.to(DEV))
    # the reference's statements (fp32, torch): the dropout mask is re-drawn from the same seed on the same device
    m = FakeModel("cpu")
    e_u, e_c = m.apply_model(x, None, uc), m.apply_model(x, None, c)
    e_t = e_u + scale * (e_c - e_u)
    assert float((cor.seen.cpu() - e_t).abs().max()) <= 1e-6 * float(e_t.abs().max()) + 1e-6
    e_t = e_t * 1.2 - 0.1 * x
    full = lambda v: torch.full((B, 1, 1, 1), float(v))  # noqa: E731
    a_t, a_prev, sig, s1m = full(s.ddim_alphas[index]), full(s.ddim_alphas_prev[index]), full(s.ddim_sigmas[index]), full(s.ddim_sqrt_one_minus_alphas[index])
    px0 = (x - s1m * e_t) / a_t.sqrt()
    torch.manual_seed(5)
    dn = torch.nn.functional.dropout(noise.to(DEV), p=p_drop).cpu()
    ref = a_prev.sqrt() * px0 + (1.0 - a_prev - sig ** 2).sqrt() * e_t + sig * dn * temp
    assert float((pred_x0.cpu() - px0).abs().max()) <= 1e-5 * float(px0.abs().max())
    assert float((x_prev.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((dn == 0).float().mean()) > 0.1, "the dropout must have dropped something"


def test_feed_forward_without_glu_and_ungated_block():
    """attention.py:61-76 with glu=False (BasicTransformerBlock(gated_ff=False)): Linear -> exact-erf GELU -> Linear, the GELU in the projection GEMM's epilogue;
    state-dict keys net.0.0.* / net.2.* as the reference's nn.Sequential gives them."""
    from anyedit_amd.ldm.modules.attention import FeedForward, BasicTransformerBlock
    torch.manual_seed(3)
    ff = FeedForward(64, glu=False)
    assert sorted(ff.state_dict().keys()) == ["net.0.0.bias", "net.0.0.weight", "net.2.bias", "net.2.weight"]
    x = torch.randn(2, 10, 64)
    qb = lambda t_: t_.to(torch.bfloat16).float()  # noqa: E731
    ref = F.linear(F.gelu(F.linear(qb(x), qb(ff.net[0][0].weight), ff.net[0][0].bias)), qb(ff.net[2].weight), ff.net[2].bias)
    got = ff.to(DEV)(x.to(DEV)).float().cpu()
    assert rel_l2(got, ref) <= 6e-3
    blk = BasicTransformerBlock(64, 2, 32, context_dim=16, gated_ff=False).to(DEV)
    y = blk(x.to(DEV), context=torch.randn(2, 5, 16).to(DEV))
    assert y.shape == x.shape and torch.isfinite(y).all()


# ------------------------------------------------------------------------------------------------------------ first stage (N1)
def test_first_stage_resampling_without_conv():
    """diffusionmodules/model.py:44-89 with with_conv=False (no kl-f8 config uses it; the reference implements it): Upsample = F.interpolate(nearest x2),
    Downsample = avg_pool2d(2, 2) — the reference's own statements, on channel counts that need the 8-channel padding too."""
    from anyedit_amd.ldm.modules.diffusionmodules.model import Upsample, Downsample
    g = torch.Generator().manual_seed(11)
    for C in (4, 32):
        x = torch.randn(2, C, 6, 10, generator=g).bfloat16().float()
        up, dn = Upsample(C, False), Downsample(C, False)
        assert len(up.state_dict()) == 0 and len(dn.state_dict()) == 0
        assert torch.equal(up(x.to(DEV)).cpu(), F.interpolate(x, scale_factor=2.0, mode="nearest"))
        close(dn(x.to(DEV)), F.avg_pool2d(x, kernel_size=2, stride=2), rl2=4e-3, what="Downsample(with_conv=False)")


def test_first_stage_autoencoder_golden():
    """AutoencoderKL encode / decode and its blocks on the HIP path vs the reference's outputs (tests/golden/vae_tiny.npz)."""
    from anyedit_amd.ldm.models.autoencoder import AutoencoderKL
    from anyedit_amd.ldm.modules.diffusionmodules import model as M
    g = load_golden("vae_tiny")
    cfg = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1,
               attn_resolutions=[], dropout=0.0)
    vae = AutoencoderKL(ddconfig=cfg, lossconfig=None, embed_dim=4)
    vae.load_state_dict(sub_sd(g, "w."))
    vae = vae.to(DEV).eval()
    x = T(g["x"]).to(DEV)
    post = vae.encode(x)
    close(post.mean, g["enc.mean"], rl2=2e-2, db=34.0, what="VAE encode mean")
    close(post.std, g["enc.std"], rl2=2e-2, db=34.0, what="VAE encode std")
    post.randn = lambda shape, device=None: T(g["sample.noise"]).to(device)
    close(post.sample(), g["sample.z"], rl2=2e-2, db=34.0, what="posterior sample")
    close(vae.decode(T(g["enc.mean"]).to(DEV)), g["dec.y"], rl2=3e-2, db=40.0, what="VAE decode")  # 20 bf16 layers deep
    h = T(g["blk.h"]).to(DEV)
    close(vae.decoder.mid.attn_1(h), g["blk.attn"], what="AttnBlock")
    close(vae.decoder.mid.block_1(h), g["blk.res"], what="ResnetBlock")
    d, u = M.Downsample(64, True), M.Upsample(64, True)
    d.load_state_dict(sub_sd(g, "down."))
    u.load_state_dict(sub_sd(g, "up."))
    close(d.to(DEV)(h), g["blk.down"], what="Downsample (asymmetric pad, stride 2)")
    close(u.to(DEV)(h), g["blk.up"], what="Upsample (nearest x2 + conv)")


def test_softmax_rows_and_gaussian_moments():
    from anyedit_amd import ops
    g = torch.Generator().manual_seed(3)
    S = torch.randn(70, 333, generator=g) * 4
    ref = torch.softmax(S * 0.125, -1)
    got = ops.softmax_rows(S.to(DEV), 0.125).float().cpu()
    assert rel_l2(got, ref) < 4e-3
    mom = torch.randn(2, 8, 4, 4, generator=g) * 20
    noise = torch.randn(2, 4, 4, 4, generator=g)
    mean, logvar = mom.chunk(2, 1)
    logvar = logvar.clamp(-30, 20)
    z, m, lv, sd = ops.gaussian_moments(mom.to(DEV), noise.to(DEV), want_stats=True)
    assert torch.equal(m.cpu(), mean) and torch.equal(lv.cpu(), logvar)
    assert rel_l2(sd.cpu(), torch.exp(0.5 * logvar)) < 1e-6 and rel_l2(z.cpu(), mean + torch.exp(0.5 * logvar) * noise) < 1e-6
    # DiagonalGaussianDistribution over it (distributions.py:24-62): sample with a replayed noise stream, mode, KL to N(0, I) and to
    # another posterior, negative log-likelihood, and the deterministic variant — against the formulas written out in torch
    from anyedit_amd.ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    mom2 = torch.randn(2, 8, 4, 4, generator=g)
    p, q = DiagonalGaussianDistribution((mom * 0.1).to(DEV)), DiagonalGaussianDistribution(mom2.to(DEV))
    p.randn = lambda shape, device=None: noise.to(device)
    m1, lv1 = (mom * 0.1).chunk(2, 1)
    m2, lv2 = mom2.chunk(2, 1)
    assert torch.equal(p.mode().cpu(), m1) and rel_l2(p.sample().cpu(), m1 + torch.exp(0.5 * lv1) * noise) < 1e-6
    assert rel_l2(p.kl().cpu(), 0.5 * (m1 ** 2 + lv1.exp() - 1 - lv1).sum((1, 2, 3))) < 1e-5
    assert rel_l2(p.kl(q).cpu(), 0.5 * ((m1 - m2) ** 2 / lv2.exp() + lv1.exp() / lv2.exp() - 1 - lv1 + lv2).sum((1, 2, 3))) < 1e-5
    xs = torch.randn(2, 4, 4, 4, generator=g)
    nll = 0.5 * (np.log(2 * np.pi) + lv1 + (xs - m1) ** 2 / lv1.exp()).sum((1, 2, 3))
    assert rel_l2(p.nll(xs.to(DEV)).cpu(), nll) < 1e-5
    det = DiagonalGaussianDistribution(mom2.to(DEV), deterministic=True)
    assert torch.equal(det.sample().cpu(), m2) and float(det.std.abs().max()) == 0.0 and float(det.kl()) == 0.0 and float(det.nll(xs)) == 0.0


def test_latent_diffusion_first_stage_roundtrip(tiny_unet):
    """LatentDiffusion.encode_first_stage / get_first_stage_encoding / decode_first_stage (ddpm.py:655-662, 822-834) wired to the
    HIP AutoencoderKL through the reference's config reflection (`ldm.models.autoencoder.AutoencoderKL` resolves in-package)."""
    unet, _ = tiny_unet
    g = load_golden("vae_tiny")
    cfg = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=1,
               attn_resolutions=[], dropout=0.0)
    ldm = _tiny_ldm(unet)
    ldm.scale_factor = 0.18215
    ldm.instantiate_first_stage({"target": "ldm.models.autoencoder.AutoencoderKL", "params": {"ddconfig": cfg, "embed_dim": 4}})
    ldm.first_stage_model.load_state_dict(sub_sd(g, "w."))
    ldm.first_stage_model.to(DEV)
    x = T(g["x"]).to(DEV)
    post = ldm.encode_first_stage(x)
    post.randn = lambda shape, device=None: T(g["sample.noise"]).to(device)
    z = ldm.get_first_stage_encoding(post)
    close(z, 0.18215 * T(g["sample.z"]), rl2=2e-2, db=34.0, what="scaled first-stage encoding")
    y = ldm.decode_first_stage(0.18215 * T(g["enc.mean"]).to(DEV))
    close(y, g["dec.y"], rl2=3e-2, db=40.0, what="decode_first_stage")


# ------------------------------------------------------------------------------------------------------------ ControlNet (N4)
def test_controlnet_and_controlled_unet_golden():
    """ControlNet / ControlledUnetModel on the HIP path vs the reference's cldm module (golden), incl. per-residual scales and
    only_mid_control."""
    from anyedit_amd.cldm.cldm import ControlNet, ControlledUnetModel
    from util_models import TINY_UNET
    g = load_golden("cldm_tiny")
    cfg = dict(TINY_UNET)
    cfg["in_channels"] = 4
    unet = ControlledUnetModel(**cfg)
    unet.load_state_dict(sub_sd(g, "unet."))
    cnet = ControlNet(hint_channels=3, **{k: v for k, v in cfg.items() if k != "out_channels"})
    cnet.load_state_dict(sub_sd(g, "cnet."))
    unet, cnet = unet.to(DEV).eval(), cnet.to(DEV).eval()
    d = lambda k: T(g[k]).to(DEV)
    with torch.no_grad():
        control = cnet(d("x"), d("hint"), d("t"), d("ctx"))
        assert len(control) == int(g["n_control"])
        for i, c in enumerate(control):
            close(c, g[f"control.{i}"], rl2=4e-2, db=40.0, what=f"control residual {i}")  # deepest one: whole bf16 encoder + 8-conv hint stack
        ctx_rows = unet.context_rows(d("ctx"))
        rows = cnet.forward_rows(d("x"), d("hint"), d("t"), ctx_rows)
        scaled = [(c, float(s_)) for c, s_ in zip(rows, g["scales"])]
        close(unet.forward_rows(d("x"), d("t"), ctx_rows, control=scaled), g["eps_control"], rl2=2.5e-2, db=36.0, what="controlled UNet")
        close(unet.forward_rows(d("x"), d("t"), ctx_rows, control=rows, only_mid_control=True), g["eps_mid_only"], rl2=2.5e-2, db=36.0,
              what="controlled UNet (mid only)")
        close(unet(d("x"), d("t"), context=d("ctx")), g["eps_plain"], rl2=2.5e-2, db=36.0, what="controlled UNet without control")
