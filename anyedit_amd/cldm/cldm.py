"""ControlNet / AnyDoor wrappers on the HIP path — mirror of AnyEdit_Collection/other_modules/cldm/cldm.py:
`ControlledUnetModel` (:21-44), `ControlNet` (:47-304), `ControlLDM.apply_model` (:328-340).  The second in-tree consumer of the
UNet operators: the control branch IS the UNet's encoder half (same ResBlocks / SpatialTransformers, same kernels) fed by a small
conv stack on the hint image, and its 13 outputs are added to the UNet's skip connections (`ae_axpy_bf16`: skip + scale*control).
State-dict layouts equal the reference's, so ControlNet / AnyDoor checkpoints load unchanged.
"""
import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel, TimestepEmbedSequential, TimestepBlock, Feat, Downsample, Upsample
from anyedit_amd.ldm.modules.diffusionmodules.util import conv_nd, zero_module
from anyedit_amd.ldm.models.diffusion.ddpm import LatentDiffusion
from anyedit_amd.ldm.util import instantiate_from_config

BF16 = torch.bfloat16


class ControlledUnetModel(UNetModel):
    def forward_rows(self, x, timesteps, context_rows, kv_cache=None, control=None, only_mid_control=False):
        """cldm.py:22-44 on channels-last rows.  control: list of bf16 row tensors (ControlNet.forward_rows order: one per skip, the
        middle one last), already multiplied by their scales, or (tensor, scale) pairs."""
        B, C, H, W = x.shape
        t_emb = ops.timestep_embedding(timesteps, self.model_channels)
        emb = self.time_embed[0].rows(t_emb, epilogue=ops.EPI_SILU)
        emb_silu = self._emb_pack(self.time_embed[2].rows(emb, epilogue=ops.EPI_SILU))
        f = Feat(ops.nchw_to_rows(x, (C + 7) // 8 * 8), B, H, W)
        hs = []
        for module in self.input_blocks:
            if isinstance(module[0], TimestepBlock) or isinstance(module[0], (Downsample, Upsample)) or len(module) > 1:
                f = module.rows(f, emb_silu, context_rows, kv_cache)
            else:
                y, _, _ = module[0].rows(f.t, B, H, W)
                f = Feat(y, B, H, W)
            hs.append(f)
        f = self.middle_block.rows(f, emb_silu, context_rows, kv_cache)
        control = None if control is None else list(control)

        def inject(t, c):
            c, s = c if isinstance(c, tuple) else (c, 1.0)
            return ops.axpy(t, c, s)

        if control is not None:
            f = Feat(inject(f.materialize(), control.pop()), f.B, f.H, f.W)
        for module in self.output_blocks:
            skip = hs.pop().materialize()
            if not (only_mid_control or control is None):
                skip = inject(skip, control.pop())
            f = Feat(f.materialize(), f.B, f.H, f.W, t2=skip)
            f = module.rows(f, emb_silu, context_rows, kv_cache)
        h = self.out[0].rows(f.materialize(), f.B, f.H * f.W, silu=True)
        y, _, _ = self.out[2].rows(h, f.B, f.H, f.W, out_f32=True)
        return ops.rows_to_nchw(y, f.B, f.H, f.W, out_dtype=torch.float32)

    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        ctx = self.context_rows(context) if context is not None else None
        ctl = None if control is None else [ops.nchw_to_rows(c) for c in control]
        out = self.forward_rows(x, timesteps, ctx, control=ctl, only_mid_control=only_mid_control)
        return out.to(x.dtype) if x.dtype != torch.float32 else out


class ControlNet(UNetModel):
    """Same constructor as the reference's ControlNet (cldm.py:48-96) = UNetModel's arguments without `out_channels`, plus
    `hint_channels`.  Built on UNetModel so the encoder half and the time embedding are literally the same modules."""

    def __init__(self, image_size, in_channels, model_channels, hint_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, use_checkpoint=False, use_fp16=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None,
                 legacy=True, disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__(image_size=image_size, in_channels=in_channels, model_channels=model_channels, out_channels=in_channels,
                         num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, dropout=dropout,
                         channel_mult=channel_mult, conv_resample=conv_resample, dims=dims, use_checkpoint=use_checkpoint,
                         use_fp16=use_fp16, num_heads=num_heads, num_head_channels=num_head_channels,
                         num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                         resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order,
                         use_spatial_transformer=use_spatial_transformer, transformer_depth=transformer_depth, context_dim=context_dim,
                         n_embed=n_embed, legacy=legacy, disable_self_attentions=disable_self_attentions,
                         num_attention_blocks=num_attention_blocks, disable_middle_self_attn=disable_middle_self_attn,
                         use_linear_in_transformer=use_linear_in_transformer)
        del self.output_blocks, self.out          # the control branch has no decoder (cldm.py:139-277)
        self.dims = dims
        self.hint_channels = hint_channels
        chans = []
        for blk in self.input_blocks:             # output width of every encoder block = width of its zero conv
            last = [m for m in blk if hasattr(m, "out_channels")][-1] if any(hasattr(m, "out_channels") for m in blk) else None
            chans.append(last.out_channels if last is not None else chans[-1])
        self.zero_convs = nn.ModuleList([self.make_zero_conv(c) for c in chans])
        self.input_hint_block = TimestepEmbedSequential(
            conv_nd(dims, hint_channels, 16, 3, padding=1), nn.SiLU(),
            conv_nd(dims, 16, 16, 3, padding=1), nn.SiLU(),
            conv_nd(dims, 16, 32, 3, padding=1, stride=2), nn.SiLU(),
            conv_nd(dims, 32, 32, 3, padding=1), nn.SiLU(),
            conv_nd(dims, 32, 96, 3, padding=1, stride=2), nn.SiLU(),
            conv_nd(dims, 96, 96, 3, padding=1), nn.SiLU(),
            conv_nd(dims, 96, 256, 3, padding=1, stride=2), nn.SiLU(),
            zero_module(conv_nd(dims, 256, model_channels, 3, padding=1)))
        self.middle_block_out = self.make_zero_conv(chans[-1])

    def make_zero_conv(self, channels):
        return TimestepEmbedSequential(zero_module(conv_nd(self.dims if hasattr(self, "dims") else 2, channels, channels, 1, padding=0)))

    def hint_rows(self, hint):
        """cldm.py:146-163: conv / SiLU stack on the hint image, three stride-2 stages (8x down, to the latent resolution)."""
        B, C, H, W = hint.shape
        h = ops.nchw_to_rows(hint, (C + 7) // 8 * 8)
        for layer in self.input_hint_block:
            if isinstance(layer, nn.SiLU):
                h = ops.silu_to_bf16(h)
            else:
                h, H, W = layer.rows(h, B, H, W)
        return h, H, W

    def forward_rows(self, x, hint, timesteps, context_rows, kv_cache=None):
        """cldm.py:283-304.  Returns the list of control rows (bf16), middle-block output last."""
        B = x.shape[0]
        t_emb = ops.timestep_embedding(timesteps, self.model_channels)
        emb = self.time_embed[0].rows(t_emb, epilogue=ops.EPI_SILU)
        emb_silu = self._emb_pack(self.time_embed[2].rows(emb, epilogue=ops.EPI_SILU))
        guided, H, W = self.hint_rows(hint)
        if (H, W) != tuple(x.shape[2:]):
            raise ValueError(f"hint resolution {hint.shape[2:]} must be 8x the latent resolution {tuple(x.shape[2:])}")
        f = Feat(guided, B, H, W)                 # "skip the first layer": the stem conv of x is never used (cldm.py:294-297)
        outs = []
        for i, (module, zero_conv) in enumerate(zip(self.input_blocks, self.zero_convs)):
            if i > 0:
                f = module.rows(f, emb_silu, context_rows, kv_cache)
            y, _, _ = zero_conv[0].rows(f.materialize(), f.B, f.H, f.W)
            outs.append(y)
        f = self.middle_block.rows(f, emb_silu, context_rows, kv_cache)
        y, _, _ = self.middle_block_out[0].rows(f.materialize(), f.B, f.H, f.W)
        outs.append(y)
        return outs

    def forward(self, x, hint, timesteps, context, **kwargs):
        outs = self.forward_rows(x, hint, timesteps, self.context_rows(context))
        shapes = []
        H, W = x.shape[2], x.shape[3]
        res = []
        for y in outs:
            hw = y.shape[0] // x.shape[0]
            side = int(round((hw * H / W) ** 0.5))
            res.append(ops.rows_to_nchw(y, x.shape[0], side, hw // side, out_dtype=torch.float32))
        return res


class ControlLDM(LatentDiffusion):
    """cldm.py:307-340 (inference surface)."""

    def __init__(self, control_stage_config, control_key=None, only_mid_control=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.control_model = control_stage_config if isinstance(control_stage_config, nn.Module) else instantiate_from_config(control_stage_config)
        self.control_key = control_key
        self.only_mid_control = only_mid_control
        self.control_scales = [1.0] * 13

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, *args, **kwargs):
        assert isinstance(cond, dict)
        unet = self.model.diffusion_model
        cond_txt = torch.cat(cond['c_crossattn'], 1)
        ctx_rows = unet.context_rows(cond_txt)
        if cond['c_concat'] is None:
            return unet.forward_rows(x_noisy, t, ctx_rows, control=None, only_mid_control=self.only_mid_control)
        control = self.control_model.forward_rows(x_noisy, torch.cat(cond['c_concat'], 1), t, ctx_rows)
        control = [(c, s) for c, s in zip(control, self.control_scales)]
        return unet.forward_rows(x_noisy, t, ctx_rows, control=control, only_mid_control=self.only_mid_control)
