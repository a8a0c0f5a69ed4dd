"""Key layouts of the weights AnyEdit loads, and the mapping between them — SURVEY.md §8(f) N4 "on-disk formats".

The reference reaches the same SD-1.5-shaped UNet through two libraries with two state-dict schemas:
  * the in-tree `ldm` UNet (`input_blocks.N.M...`, ldm/modules/diffusionmodules/openaimodel.py:413-786) — what this package mirrors and
    what `.ckpt` files of the CompVis lineage hold under `model.diffusion_model.`;
  * diffusers' UNet2DConditionModel / AutoencoderKL (`down_blocks.i.resnets.j...`), which is what `from_pretrained(..., subfolder="unet")`
    reads in train.py:405-412 and tools/global_tool.py:74-76 (InstructPix2Pix / AnySD weights ship in this layout).
The mapping is DERIVED by walking the constructed module (which block of which level is a ResBlock / SpatialTransformer / Down- or
Upsample), not from a table, so it holds for every geometry the UNet mirror builds (SD-1.5, SD-2.1 / AnyDoor, the tiny test nets).
diffusers is a third-party dependency absent from /root/reference and from this image: its side of the mapping is restated from its
published schema — PARITY UNPINNED (tests check bijectivity, shapes and a set of well-known key pairs).
"""
import re

from anyedit_amd.ldm.modules.attention import SpatialTransformer
from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample

_RES = (("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"), ("out_layers.0", "norm2"),
        ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut"))


def unet_prefix_map(unet):
    """ldm module-path prefix -> diffusers prefix, for every leaf group of `unet` (a UNetModel mirror)."""
    pm = {"time_embed.0": "time_embedding.linear_1", "time_embed.2": "time_embedding.linear_2", "input_blocks.0.0": "conv_in",
          "out.0": "conv_norm_out", "out.2": "conv_out"}

    def add_res(ldm, dif):
        for a, b in _RES:
            pm[f"{ldm}.{a}"] = f"{dif}.{b}"

    level, n_res, n_att = 0, 0, 0
    for i, blk in enumerate(unet.input_blocks):
        if i == 0:
            continue
        for j, m in enumerate(blk):
            if isinstance(m, ResBlock):
                add_res(f"input_blocks.{i}.{j}", f"down_blocks.{level}.resnets.{n_res}")
                n_res += 1
            elif isinstance(m, SpatialTransformer):
                pm[f"input_blocks.{i}.{j}"] = f"down_blocks.{level}.attentions.{n_att}"
                n_att += 1
            elif isinstance(m, Downsample):
                pm[f"input_blocks.{i}.{j}.op"] = f"down_blocks.{level}.downsamplers.0.conv"
                level, n_res, n_att = level + 1, 0, 0
    n_res = n_att = 0
    for j, m in enumerate(unet.middle_block):
        if isinstance(m, ResBlock):
            add_res(f"middle_block.{j}", f"mid_block.resnets.{n_res}")
            n_res += 1
        elif isinstance(m, SpatialTransformer):
            pm[f"middle_block.{j}"] = f"mid_block.attentions.{n_att}"
            n_att += 1
    level, n_res, n_att = 0, 0, 0
    for i, blk in enumerate(unet.output_blocks):
        for j, m in enumerate(blk):
            if isinstance(m, ResBlock):
                add_res(f"output_blocks.{i}.{j}", f"up_blocks.{level}.resnets.{n_res}")
                n_res += 1
            elif isinstance(m, SpatialTransformer):
                pm[f"output_blocks.{i}.{j}"] = f"up_blocks.{level}.attentions.{n_att}"
                n_att += 1
            elif isinstance(m, Upsample):
                pm[f"output_blocks.{i}.{j}.conv"] = f"up_blocks.{level}.upsamplers.0.conv"
                level, n_res, n_att = level + 1, 0, 0
    return pm


def _key_map(keys, prefix_map):
    """Longest-prefix rewrite of every key; raises on a key no prefix covers."""
    order = sorted(prefix_map, key=len, reverse=True)
    out = {}
    for k in keys:
        for p in order:
            if k == p or k.startswith(p + "."):
                out[k] = prefix_map[p] + k[len(p):]
                break
        else:
            raise KeyError(f"no layout rule covers key '{k}'")
    return out


def unet_ldm_to_diffusers_keys(unet):
    """{ldm key: diffusers key} for every entry of unet.state_dict()."""
    return _key_map(unet.state_dict().keys(), unet_prefix_map(unet))


def _fit(t, like):
    """1x1-conv <-> linear weights differ only by trailing singleton dims between the two libraries' variants."""
    if t.shape != like.shape and t.numel() == like.numel() and t.squeeze().shape == like.squeeze().shape:
        return t.reshape(like.shape)
    return t


def convert_diffusers_unet(unet, diffusers_sd):
    """diffusers UNet2DConditionModel state dict -> state dict in `unet`'s (ldm) layout.  Every key must be consumed and every
    parameter of `unet` produced, with equal shapes (up to the 1x1-conv/linear reshape); anything else raises."""
    k2d = unet_ldm_to_diffusers_keys(unet)
    ref = unet.state_dict()
    missing = [d for d in k2d.values() if d not in diffusers_sd]
    extra = sorted(set(diffusers_sd) - set(k2d.values()))
    if missing or extra:
        raise KeyError(f"diffusers UNet layout mismatch: {len(missing)} missing (e.g. {missing[:3]}), {len(extra)} unexpected (e.g. {extra[:3]})")
    out = {}
    for k, d in k2d.items():
        t = _fit(diffusers_sd[d], ref[k])
        if t.shape != ref[k].shape:
            raise ValueError(f"{d} -> {k}: shape {tuple(t.shape)} != {tuple(ref[k].shape)}")
        out[k] = t
    return out


def convert_unet_to_diffusers(unet, ldm_sd=None):
    """The inverse direction (e.g. to hand trained weights back to a diffusers pipeline)."""
    sd = unet.state_dict() if ldm_sd is None else ldm_sd
    return {d: sd[k] for k, d in unet_ldm_to_diffusers_keys(unet).items()}


# ------------------------------------------------------------------------------------------------------------------ VAE
_VAE_RES = (("nin_shortcut", "conv_shortcut"),)
_VAE_ATTN = (("norm", "group_norm"), ("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0"))


def vae_prefix_map(vae):
    """ldm AutoencoderKL (ldm/modules/diffusionmodules/model.py:452-653) prefix -> diffusers AutoencoderKL prefix."""
    pm = {"quant_conv": "quant_conv", "post_quant_conv": "post_quant_conv"}
    for side in ("encoder", "decoder"):
        net = getattr(vae, side)
        for name in ("conv_in", "conv_out"):
            pm[f"{side}.{name}"] = f"{side}.{name}"
        pm[f"{side}.norm_out"] = f"{side}.conv_norm_out"
        pm[f"{side}.mid.block_1"] = f"{side}.mid_block.resnets.0"
        pm[f"{side}.mid.block_2"] = f"{side}.mid_block.resnets.1"
        for a, b in _VAE_ATTN:
            pm[f"{side}.mid.attn_1.{a}"] = f"{side}.mid_block.attentions.0.{b}"
        levels = net.down if side == "encoder" else net.up
        n = len(levels)
        for i, lv in enumerate(levels):
            di = i if side == "encoder" else n - 1 - i                 # the ldm decoder indexes `up` from the lowest resolution's end
            grp = "down_blocks" if side == "encoder" else "up_blocks"
            for j in range(len(lv.block)):
                pm[f"{side}.{'down' if side == 'encoder' else 'up'}.{i}.block.{j}"] = f"{side}.{grp}.{di}.resnets.{j}"
            if hasattr(lv, "downsample"):
                pm[f"{side}.down.{i}.downsample.conv"] = f"{side}.down_blocks.{di}.downsamplers.0.conv"
            if hasattr(lv, "upsample"):
                pm[f"{side}.up.{i}.upsample.conv"] = f"{side}.up_blocks.{di}.upsamplers.0.conv"
    return pm


def vae_ldm_to_diffusers_keys(vae):
    km = _key_map([k for k in vae.state_dict().keys() if not k.startswith("loss.")], vae_prefix_map(vae))
    return {k: re.sub(r"\.nin_shortcut\.", ".conv_shortcut.", d) for k, d in km.items()}


def convert_diffusers_vae(vae, diffusers_sd):
    """diffusers AutoencoderKL state dict -> `vae`'s (ldm) layout; the mid-block attention projections are Linear there and 1x1 convs
    here (reshape only).  Older diffusers files name them query/key/value/proj_attn: accepted as aliases."""
    alias = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    ref = vae.state_dict()
    out = {}
    for k, d in vae_ldm_to_diffusers_keys(vae).items():
        if d not in diffusers_sd:
            for new, old in alias.items():
                if f".attentions.0.{new}." in d and d.replace(f".{new}.", f".{old}.") in diffusers_sd:
                    d = d.replace(f".{new}.", f".{old}.")
                    break
            else:
                raise KeyError(f"diffusers VAE layout mismatch: '{d}' (for '{k}') not found")
        t = _fit(diffusers_sd[d], ref[k])
        if t.shape != ref[k].shape:
            raise ValueError(f"{d} -> {k}: shape {tuple(t.shape)} != {tuple(ref[k].shape)}")
        out[k] = t
    return out


def load_unet_weights(unet, path, location="cpu"):
    """Load a UNet from any of the on-disk forms: an ldm state dict (bare, or under `model.diffusion_model.` in a full SD checkpoint)
    or a diffusers `diffusion_pytorch_model.{safetensors,bin}`.  Returns the layout that was recognised."""
    from anyedit_amd.cldm.model import load_state_dict
    sd = load_state_dict(path, location)
    own = set(unet.state_dict().keys())
    if own <= set(sd.keys()):
        unet.load_state_dict({k: sd[k] for k in own})
        return "ldm"
    pref = "model.diffusion_model."
    if all(pref + k in sd for k in own):
        unet.load_state_dict({k: sd[pref + k] for k in own})
        return "ldm-checkpoint"
    unet.load_state_dict(convert_diffusers_unet(unet, sd))
    return "diffusers"
