"""Oracle (test infrastructure) for row A9 — PARITY UNPINNED.

The reference's AnySD package is absent (empty submodule github.com/weichow23/AnySD, branch main, no commit pin; call sites
train.py:25-28, 410-424, 483-485, 694-695; no reference test pins it).  This file restates OUR spec (anyedit_amd/anysd/model.py
docstring) in fp32 on the CPU so the HIP path can at least be checked for self-consistency.
"""
import torch
import torch.nn.functional as F

from . import ldm_ref as L


def task_gate(task_embs, edit_code, Wg, bg):
    te = task_embs[edit_code.long()]
    probs = torch.softmax(F.linear(te, Wg, bg), dim=-1)
    top1p, top1 = probs.max(dim=-1)  # torch.max returns the first maximal index = lowest-index tie-break
    return probs, top1, top1p


def image_proj(sd, ref_embeds, tokens, cross_dim):
    y = F.linear(ref_embeds[:, 0], sd["image_proj_model.proj.weight"], sd["image_proj_model.proj.bias"])
    y = y.reshape(-1, tokens, cross_dim)
    return F.layer_norm(y, (cross_dim,), sd["image_proj_model.norm.weight"], sd["image_proj_model.norm.bias"], 1e-5)


def moe_forward(unet_sd, unet_cfg, moe_sd, attn2_prefixes, x, t, ehs, ref_embeds, edit_code, tokens=4):
    """attn2_prefixes: state-dict prefixes of the UNet's cross-attention layers in module order (= adapter_modules order)."""
    Dc = ehs.shape[-1]
    te = moe_sd["task_embs"][edit_code.long()]
    ctx = torch.cat([ehs, te[:, None, :]], dim=1)
    _, top1, top1p = task_gate(moe_sd["task_embs"], edit_code, moe_sd["gate.weight"], moe_sd["gate.bias"])
    ip = image_proj(moe_sd, ref_embeds, tokens, Dc)
    adapters = {}
    for l, p in enumerate(attn2_prefixes):
        W = moe_sd[f"adapter_modules.{l}"][top1]            # [B, 2*inner, Dc]
        kv = torch.einsum("btd,bnd->btn", ip, W)           # [B, T, 2*inner]
        inner = kv.shape[-1] // 2
        adapters[p] = (kv[..., :inner], kv[..., inner:], top1p)
    return L.unet_forward(unet_sd, unet_cfg, x, t, ctx, adapters=adapters)
