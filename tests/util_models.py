"""Test helpers: rebuild the golden-fixture models with anyedit_amd classes (same seeds / same RNG order as
tools/gen_golden.py, which ran the reference constructors)."""
import torch
import torch.nn as nn

TINY_UNET = dict(image_size=8, in_channels=8, model_channels=32, out_channels=4, num_res_blocks=1,
                 attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=4, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=16, legacy=False, use_checkpoint=False)


def G(seed):
    return torch.Generator().manual_seed(seed)


def unzero(module, gen, std=0.02):
    from anyedit_amd.ldm.modules.diffusionmodules import openaimodel as om
    from anyedit_amd.ldm.modules import attention as at
    for m in module.modules():
        targets = []
        if isinstance(m, om.ResBlock):
            targets.append(m.out_layers[-1])
        if isinstance(m, at.SpatialTransformer):
            targets.append(m.proj_out)
        if isinstance(m, om.UNetModel):
            targets.append(m.out[-1])
        for t in targets:
            for p in t.parameters():
                p.data = torch.randn(p.shape, generator=gen) * std


def randomize_norm_affine(module, gen):
    for m in module.modules():
        if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
            m.weight.data = 1.0 + 0.1 * torch.randn(m.weight.shape, generator=gen)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=gen)


def build_tiny_unet(seed=50):
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.manual_seed(seed)
    g = G(seed)
    unet = UNetModel(**TINY_UNET)
    unzero(unet, g, std=0.05)
    randomize_norm_affine(unet, g)
    return unet.eval()
