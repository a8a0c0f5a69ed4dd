"""Oracle (test infrastructure): fp32 CPU restatement of the ldm UNet hot path, functional over a
reference-layout state dict (keys as in SURVEY.md Appendix A).

Rows: A1/A2 CrossAttention (ldm/modules/attention.py:145-194), A3 BasicTransformerBlock/GEGLU
(:49-76, :271-275), A4 SpatialTransformer (:321-340), A5 ResBlock
(ldm/modules/diffusionmodules/openaimodel.py:254-274), Down/Upsample (:108-118,:157-159),
A6 GroupNorm32 (util.py:217-219) / Normalize (attention.py:88-89), A7 UNetModel.forward
(openaimodel.py:754-786; constructor walk :542-730).
"""
import torch
import torch.nn.functional as F

import contextlib
import math
import threading

from .schedule_ref import timestep_embedding

# ------------------------------------------------------------------ storage-precision control
# `_st` marks the points where the HIP path STORES an activation (bf16 rows in HBM).  It is the identity by default, so the
# oracle stays the reference's fp32 arithmetic bit for bit.  Inside `bf16_storage()` it rounds to bfloat16 and back: the
# oracle then carries the storage error any bf16-activation implementation of this graph must have, with fp32 arithmetic
# everywhere else.  Tests use it as the CONTROL for the tolerance: HIP error vs fp32 oracle <= 1.5 x control error.
_tls = threading.local()  # per thread: a test may run the fp32 oracle and the control side by side


def _st(x):
    r = getattr(_tls, "round", None)
    return x if r is None else r(x)


@contextlib.contextmanager
def bf16_storage():
    prev = getattr(_tls, "round", None)
    _tls.round = lambda t: t.to(torch.bfloat16).to(torch.float32)
    try:
        yield
    finally:
        _tls.round = prev


def bf16_weights(sd):
    """State dict with every floating tensor rounded to bfloat16 (what `ops.pack_*` stores), kept as fp32 for the CPU ops."""
    return {k: (v.to(torch.bfloat16).to(torch.float32) if torch.is_floating_point(v) else v) for k, v in sd.items()}


# ------------------------------------------------------------------ A6 norms
def group_norm32(x, w, b, eps=1e-5, groups=32):
    """util.py:217-219: GroupNorm computed in fp32, cast back to the input dtype."""
    return F.group_norm(x.float(), groups, w.float(), b.float(), eps).type(x.dtype)


def group_norm_nhwc_manual(x, w, b, eps, groups=32):
    """Independent (non-ATen) statement of the same arithmetic on [B,HW,C]; used to cross-check."""
    B, N, C = x.shape
    xg = x.float().reshape(B, N, groups, C // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).reshape(B, N, C)
    return y * w + b


def silu(x):
    return x * torch.sigmoid(x)


# ------------------------------------------------------------------ A1/A2 attention
def cross_attention(sd, p, x, context=None, mask=None, heads=8, adapter=None):
    """attention.py:163-194.  sd[p+'to_q.weight'] etc.  fp32 logits, softmax(-1), PV, to_out.
    adapter=(k_ip [B,T,inner], v_ip [B,T,inner], gate [B]): OUR AnySD spec (row A9, parity unpinned): a decoupled second
    attention over expert K/V added before to_out, as ip_adapter/attention_processor.py:141-173 does."""
    q = _st(F.linear(x, sd[p + "to_q.weight"]))
    ctx = x if context is None else context
    k = _st(F.linear(ctx, sd[p + "to_k.weight"]))
    v = _st(F.linear(ctx, sd[p + "to_v.weight"]))
    B, N, inner = q.shape
    d = inner // heads
    scale = d ** -0.5

    def split(t):  # 'b n (h d) -> (b h) n d'
        return t.reshape(B, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(B * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * scale
    if mask is not None:
        m = mask.reshape(B, -1)
        m = m[:, None, None, :].expand(B, heads, 1, m.shape[-1]).reshape(B * heads, 1, -1)
        sim = sim.masked_fill(~m, -torch.finfo(sim.dtype).max)
    sim = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", sim, v)
    if adapter is None:
        out = _st(out)
    if adapter is not None:
        k_ip, v_ip, gate = adapter
        k_ip, v_ip = split(k_ip), split(v_ip)
        sim_ip = (torch.einsum("bid,bjd->bij", q.float(), k_ip.float()) * scale).softmax(dim=-1)
        out = _st(out + gate.repeat_interleave(heads)[:, None, None] * torch.einsum("bij,bjd->bid", sim_ip, v_ip))
    out = out.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, inner)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def sdpa_core(q, k, v, scale, bias=None):
    """softmax(q k^T * scale + bias) v on [BH,N,D] fp32 — the fused-op contract (attention.py:222-233)."""
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * scale
    if bias is not None:
        sim = sim + bias
    return torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v.float())


# ------------------------------------------------------------------ A3 transformer block
def geglu_ff(sd, p, x):
    """attention.py:49-76: proj -> chunk -> x*gelu(gate) (exact-erf GELU) -> Linear."""
    h = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = _st(a * F.gelu(gate))
    return F.linear(h, sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def basic_transformer_block(sd, p, x, context, heads, disable_self_attn=False, adapters=None):
    """attention.py:271-275 (LayerNorm eps 1e-5, :263-265)."""
    C = x.shape[-1]
    ln = lambda t, i: _st(F.layer_norm(t, (C,), sd[p + f"norm{i}.weight"], sd[p + f"norm{i}.bias"], 1e-5))
    x = _st(cross_attention(sd, p + "attn1.", ln(x, 1), context if disable_self_attn else None, heads=heads) + x)
    x = _st(cross_attention(sd, p + "attn2.", ln(x, 2), context, heads=heads,
                            adapter=None if adapters is None else adapters.get(p + "attn2.")) + x)
    x = _st(geglu_ff(sd, p + "ff.", ln(x, 3)) + x)
    return x


# ------------------------------------------------------------------ A4 spatial transformer
def spatial_transformer(sd, p, x, context, heads, depth=1, use_linear=False, adapters=None):
    """attention.py:321-340 (GroupNorm eps 1e-6, :88-89)."""
    B, C, H, W = x.shape
    x_in = x
    x = _st(F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6))
    if not use_linear:
        x = _st(F.conv2d(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"]))
    x = x.permute(0, 2, 3, 1).reshape(B, H * W, -1)
    if use_linear:
        x = _st(F.linear(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"]))
    for d in range(depth):
        x = basic_transformer_block(sd, p + f"transformer_blocks.{d}.", x, context, heads, adapters=adapters)
    if use_linear:
        x = F.linear(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = F.conv2d(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return _st(x + x_in)


# ------------------------------------------------------------------ A5 resblock & resampling
def resblock(sd, p, x, emb, up=False, down=False, scale_shift=False):
    """openaimodel.py:254-274.  up / down (resblock_updown, :215-221): the resampling WITHOUT a conv sits between in_rest (GroupNorm + SiLU) and in_conv,
    and on x in front of the skip (:255-260); scale_shift (use_scale_shift_norm, :264-268): emb_out is (scale | shift) and h = out_norm(h) * (1 + scale) + shift."""
    h = _st(silu(group_norm32(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"])))
    if up:
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h, x = _st(F.avg_pool2d(h, 2, 2)), _st(F.avg_pool2d(x, 2, 2))
    h = F.conv2d(h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    emb_out = F.linear(silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    if scale_shift:
        scale, shift = torch.chunk(emb_out[:, :, None, None], 2, dim=1)
        h = _st(group_norm32(_st(h), sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"]))
        h = _st(silu(h * (1 + scale) + shift))
    else:
        h = _st(h + emb_out[:, :, None, None])
        h = _st(silu(group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"])))
    h = F.conv2d(h, sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if (p + "skip_connection.weight") in sd:
        wsk = sd[p + "skip_connection.weight"]
        x = _st(F.conv2d(x, wsk, sd[p + "skip_connection.bias"], padding=wsk.shape[-1] // 2))
    return _st(x + h)


def downsample(sd, p, x):
    """openaimodel.py:157-159: conv3x3 stride 2 pad 1; without a conv (conv_resample=False, :152-155) the 2x2 mean."""
    if (p + "op.weight") not in sd:
        return _st(F.avg_pool2d(x, 2, 2))
    return _st(F.conv2d(x, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1))


def upsample(sd, p, x):
    """openaimodel.py:108-118: nearest x2 then conv3x3 (no conv with conv_resample=False)."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if (p + "conv.weight") not in sd:
        return x
    return _st(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1))


def attention_block(sd, p, x, heads, new_order=False):
    """openaimodel.py:316-322 + QKVAttentionLegacy (:344-373) / QKVAttention (:376-409): GroupNorm -> pointwise qkv -> per-head softmax(q k^T / sqrt(ch)) v over
    all positions -> pointwise proj_out + residual.  Legacy order: channels (head, {q, k, v}, ch); new order: ({q, k, v}, head, ch)."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    T = xf.shape[-1]
    qkv = _st(F.conv1d(_st(group_norm32(xf, sd[p + "norm.weight"], sd[p + "norm.bias"])), sd[p + "qkv.weight"], sd[p + "qkv.bias"]))
    ch = qkv.shape[1] // (3 * heads)
    if new_order:
        q, k, v = (t.reshape(b * heads, ch, T) for t in qkv.chunk(3, dim=1))
    else:
        q, k, v = qkv.reshape(b * heads, ch * 3, T).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale).float(), dim=-1)
    a = _st(torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, T))
    h = F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return _st(xf + h).reshape(x.shape)


# ------------------------------------------------------------------ A7 UNet
def unet_plan(cfg):
    """Walk of UNetModel.__init__ (openaimodel.py:542-730) for use_spatial_transformer=True, legacy=False
    (heads fixed by num_heads, or by num_head_channels as in the SD-2.1 / AnyDoor configs).  Returns (input_blocks, middle, output_blocks): lists of layer tuples."""
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    attn_res = set(cfg["attention_resolutions"])
    nhc = cfg.get("num_head_channels", -1)
    hd = (lambda c: (cfg["num_heads"], c // cfg["num_heads"])) if nhc == -1 else (lambda c: (c // nhc, nhc))   # openaimodel.py:586-592
    depth = cfg.get("transformer_depth", 1)
    if not cfg.get("use_spatial_transformer", cfg.get("context_dim") is not None):   # (configs written before this branch existed carry a context_dim and no flag)
        # AttentionBlock layers (openaimodel.py:568-588, 640-645, 680-699): heads from num_head_channels when given, else num_heads (input / middle) or
        # num_heads_upsample (output) under `legacy`, else ch // (ch // num_heads)
        legacy, nh_in = cfg.get("legacy", True), cfg.get("num_heads", -1)
        nh_up = cfg.get("num_heads_upsample", -1)
        nh_up = nh_in if nh_up == -1 else nh_up
        new_order = bool(cfg.get("use_new_attention_order", False))

        def attn_entry(ch, side_heads):
            if nhc != -1:
                return ("attn", ch, ch // nhc, new_order)
            return ("attn", ch, side_heads if legacy else ch // (ch // nh_in), new_order)
    else:
        attn_entry = None
    inp = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb[level]):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("st", ch, *hd(ch), depth) if attn_entry is None else attn_entry(ch, nh_in))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("resdown" if cfg.get("resblock_updown") else "down", ch, ch)])   # openaimodel.py:600-616
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch), ("st", ch, *hd(ch), depth) if attn_entry is None else attn_entry(ch, nh_in), ("res", ch, ch)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb[level] + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(("st", ch, *hd(ch), depth) if attn_entry is None else attn_entry(ch, nh_up))
            if level and i == nrb[level]:
                layers.append(("resup" if cfg.get("resblock_updown") else "up", ch, ch))   # openaimodel.py:707-721
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _run_layers(sd, prefix, layers, h, emb, context, use_linear, adapters=None, scale_shift=False):
    for j, L in enumerate(layers):
        p = f"{prefix}{j}."
        kind = L[0]
        if kind == "conv":
            h = _st(F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1))
        elif kind == "res":
            h = resblock(sd, p, h, emb, scale_shift=scale_shift)
        elif kind in ("resdown", "resup"):
            h = resblock(sd, p, h, emb, up=kind == "resup", down=kind == "resdown", scale_shift=scale_shift)
        elif kind == "attn":
            h = attention_block(sd, p, h, heads=L[2], new_order=L[3])
        elif kind == "st":
            h = spatial_transformer(sd, p, h, context, heads=L[2], depth=L[4], use_linear=use_linear, adapters=adapters)
        elif kind == "down":
            h = downsample(sd, p, h)
        elif kind == "up":
            h = upsample(sd, p, h)
    return h


def unet_forward(sd, cfg, x, timesteps, context, adapters=None, y=None):
    """openaimodel.py:754-786; y: class labels [B] of a class-conditional model (num_classes an int: :533-535, 770-772) or real values
    [B, 1] for num_classes == "continuous" (a Linear(1, 4*mc) label embedding, :536-538)."""
    inp, mid, out = unet_plan(cfg)
    use_linear = cfg.get("use_linear_in_transformer", False)
    ssn = bool(cfg.get("use_scale_shift_norm", False))
    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    assert (y is not None) == (cfg.get("num_classes") is not None), "must specify y if and only if the model is class-conditional"
    if y is not None:
        if cfg["num_classes"] == "continuous":
            emb = emb + F.linear(y.float(), sd["label_emb.weight"], sd["label_emb.bias"])
        else:
            emb = emb + sd["label_emb.weight"][y.long()]
    hs = []
    h = _st(x.float())
    context = None if context is None else _st(context)
    for i, layers in enumerate(inp):
        h = _run_layers(sd, f"input_blocks.{i}.", layers, h, emb, context, use_linear, adapters, ssn)
        hs.append(h)
    h = _run_layers(sd, "middle_block.", mid, h, emb, context, use_linear, adapters, ssn)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, f"output_blocks.{i}.", layers, h, emb, context, use_linear, adapters, ssn)
    if cfg.get("n_embed") is not None:   # predict_codebook_ids (openaimodel.py:731-736, 783-784): GroupNorm + pointwise conv, no SiLU
        return F.conv2d(_st(group_norm32(h, sd["id_predictor.0.weight"], sd["id_predictor.0.bias"])), sd["id_predictor.1.weight"], sd["id_predictor.1.bias"])
    h = _st(silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"])))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def diffusion_wrapper(sd, cfg, x, t, c_concat=None, c_crossattn=None, conditioning_key="hybrid", c_adm=None):
    """ldm/models/diffusion/ddpm.py:1332-1363."""
    if conditioning_key == "hybrid-adm":
        return unet_forward(sd, cfg, torch.cat([x] + c_concat, dim=1), t, torch.cat(c_crossattn, 1), y=c_adm)
    if conditioning_key == "crossattn-adm":
        return unet_forward(sd, cfg, x, t, torch.cat(c_crossattn, 1), y=c_adm)
    if conditioning_key == "adm":
        return unet_forward(sd, cfg, x, t, None, y=c_crossattn[0])
    if conditioning_key is None:
        return unet_forward(sd, cfg, x, t, None)
    if conditioning_key == "concat":
        return unet_forward(sd, cfg, torch.cat([x] + c_concat, dim=1), t, None)
    if conditioning_key == "crossattn":
        return unet_forward(sd, cfg, x, t, torch.cat(c_crossattn, 1))
    if conditioning_key == "hybrid":
        return unet_forward(sd, cfg, torch.cat([x] + c_concat, dim=1), t, torch.cat(c_crossattn, 1))
    raise NotImplementedError(conditioning_key)


SD15_CFG = dict(image_size=64, in_channels=8, model_channels=320, out_channels=4, num_res_blocks=2,
                attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8,
                use_spatial_transformer=True, transformer_depth=1, context_dim=768, legacy=False)
