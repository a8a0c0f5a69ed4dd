"""Oracle (test infrastructure): fp32 CPU restatement of the SAM ViT image-encoder hot path (row A10).

Follows segment_anything/segment_anything/modeling/image_encoder.py:106-116 (encoder forward),
:166-182 (Block), :224-240 (Attention), :243-289 (window partition), :292-361 (decomposed rel-pos),
:364-395 (PatchEmbed) and modeling/common.py:13-43 (MLPBlock, LayerNorm2d).
Functional over a reference-layout state dict.
"""
import torch
import torch.nn.functional as F


def get_rel_pos(q_size, k_size, rel_pos):
    """image_encoder.py:292-322."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def decomposed_rel_pos_terms(q, rel_pos_h, rel_pos_w, q_size, k_size):
    """image_encoder.py:325-355: rel_h[b,qh,qw,kh], rel_w[b,qh,qw,kw] from the UNSCALED q (G13)."""
    q_h, q_w = q_size
    k_h, k_w = k_size
    Rh = get_rel_pos(q_h, k_h, rel_pos_h)
    Rw = get_rel_pos(q_w, k_w, rel_pos_w)
    B, _, dim = q.shape
    r_q = q.reshape(B, q_h, q_w, dim)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    return rel_h, rel_w


def add_decomposed_rel_pos(attn, q, rel_pos_h, rel_pos_w, q_size, k_size):
    """image_encoder.py:325-361."""
    q_h, q_w = q_size
    k_h, k_w = k_size
    rel_h, rel_w = decomposed_rel_pos_terms(q, rel_pos_h, rel_pos_w, q_size, k_size)
    B = q.shape[0]
    attn = (attn.view(B, q_h, q_w, k_h, k_w) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :])
    return attn.view(B, q_h * q_w, k_h * k_w)


def attention(sd, p, x, num_heads, use_rel_pos=True):
    """image_encoder.py:224-240."""
    B, H, W, C = x.shape
    d = C // num_heads
    scale = d ** -0.5
    qkv = F.linear(x, sd[p + "qkv.weight"], sd.get(p + "qkv.bias"))
    qkv = qkv.reshape(B, H * W, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * W, -1).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1)
    if use_rel_pos:
        attn = add_decomposed_rel_pos(attn, q, sd[p + "rel_pos_h"], sd[p + "rel_pos_w"], (H, W), (H, W))
    attn = attn.softmax(dim=-1)
    x = (attn @ v).view(B, num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def window_partition(x, ws):
    """image_encoder.py:243-264 (zero pad bottom/right; G12: padded tokens are NOT masked)."""
    B, H, W, C = x.shape
    pad_h = (ws - H % ws) % ws
    pad_w = (ws - W % ws) % ws
    if pad_h > 0 or pad_w > 0:
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))
    Hp, Wp = H + pad_h, W + pad_w
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(windows, ws, pad_hw, hw):
    """image_encoder.py:267-289."""
    Hp, Wp = pad_hw
    H, W = hw
    B = windows.shape[0] // (Hp * Wp // ws // ws)
    x = windows.view(B, Hp // ws, Wp // ws, ws, ws, -1)
    x = x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    if Hp > H or Wp > W:
        x = x[:, :H, :W, :].contiguous()
    return x


def block(sd, p, x, num_heads, window_size, ln_eps=1e-6):
    """image_encoder.py:166-182."""
    C = x.shape[-1]
    shortcut = x
    x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], ln_eps)
    if window_size > 0:
        H, W = x.shape[1], x.shape[2]
        x, pad_hw = window_partition(x, window_size)
    x = attention(sd, p + "attn.", x, num_heads)
    if window_size > 0:
        x = window_unpartition(x, window_size, pad_hw, (H, W))
    x = shortcut + x
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], ln_eps)
    h = F.linear(h, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])
    h = F.linear(F.gelu(h), sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
    return x + h


def layer_norm_2d(x, w, b, eps=1e-6):
    """common.py:30-43."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def image_encoder(sd, p, x, patch_size, depth, num_heads, window_size, global_attn_indexes):
    """image_encoder.py:106-116."""
    x = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=patch_size)
    x = x.permute(0, 2, 3, 1)
    if (p + "pos_embed") in sd:
        x = x + sd[p + "pos_embed"]
    for i in range(depth):
        ws = 0 if i in global_attn_indexes else window_size
        x = block(sd, p + f"blocks.{i}.", x, num_heads, ws)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[p + "neck.0.weight"])
    x = layer_norm_2d(x, sd[p + "neck.1.weight"], sd[p + "neck.1.bias"])
    x = F.conv2d(x, sd[p + "neck.2.weight"], padding=1)
    x = layer_norm_2d(x, sd[p + "neck.3.weight"], sd[p + "neck.3.bias"])
    return x


VIT_H = dict(img_size=1024, patch_size=16, embed_dim=1280, depth=32, num_heads=16, window_size=14,
             global_attn_indexes=(7, 15, 23, 31), out_chans=256)  # build_sam.py:14-22, 65-80
