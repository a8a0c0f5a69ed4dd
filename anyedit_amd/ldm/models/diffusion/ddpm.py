"""Mirror of the hot-path subset of ldm/models/diffusion/ddpm.py: DDPM schedule buffers (ddpm.py:138-192), q_sample
(:356-359), eps-MSE (:367-380, :889-932), LatentDiffusion.apply_model (:854-869), DiffusionWrapper (:1324-1363).
The Lightning trainer / logging / first-stage / cond-stage glue is out of scope.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.ldm.util import instantiate_from_config, default, exists
from anyedit_amd.ldm.modules.diffusionmodules.util import make_beta_schedule, extract_into_tensor


class DiffusionWrapper(nn.Module):
    """ddpm.py:1324-1363: conditioning-key switch in front of the UNet."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        if isinstance(diff_model_config, nn.Module):
            self.sequential_cross_attn = False
            self.diffusion_model = diff_model_config
        else:
            diff_model_config = dict(diff_model_config)
            self.sequential_cross_attn = diff_model_config.pop("sequential_crossattn", False)
            self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, 'concat', 'crossattn', 'hybrid', 'adm', 'hybrid-adm', 'crossattn-adm']
        self.kv_cache = None  # set by samplers: projected K|V of a step-invariant context

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, c_adm=None):
        dm = self.diffusion_model
        if self.conditioning_key is None:
            return dm(x, t)
        if self.conditioning_key == 'concat':
            return dm(torch.cat([x] + c_concat, dim=1), t)
        if self.conditioning_key == 'crossattn':
            cc = torch.cat(c_crossattn, 1) if not self.sequential_cross_attn else c_crossattn
            return self._run(x, t, cc)
        if self.conditioning_key == 'hybrid':
            xc = torch.cat([x] + c_concat, dim=1)
            cc = torch.cat(c_crossattn, 1)
            return self._run(xc, t, cc)
        if self.conditioning_key == 'hybrid-adm':       # ddpm.py:1349-1353
            assert c_adm is not None
            return self._run(torch.cat([x] + c_concat, dim=1), t, torch.cat(c_crossattn, 1), y=c_adm)
        if self.conditioning_key == 'crossattn-adm':    # :1354-1357
            assert c_adm is not None
            return self._run(x, t, torch.cat(c_crossattn, 1), y=c_adm)
        if self.conditioning_key == 'adm':              # :1358-1360
            return dm(x, t, y=c_crossattn[0])
        raise NotImplementedError()

    def _run(self, x, t, cc, y=None):
        dm = self.diffusion_model
        if hasattr(dm, "forward_rows"):
            if y is not None:
                return dm.forward_rows(x, t, dm.context_rows(cc), kv_cache=self.kv_cache, y=y)
            return dm.forward_rows(x, t, dm.context_rows(cc), kv_cache=self.kv_cache)
        return dm(x, t, context=cc) if y is None else dm(x, t, context=cc, y=y)


class DDPM(nn.Module):
    """Schedule owner (ddpm.py:46-192, 356-411): the `model` duck type DDIMSampler needs."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, v_posterior=0., parameterization="eps", conditioning_key=None,
                 loss_type="l2", image_size=64, channels=4, **ignored):
        super().__init__()
        assert parameterization in ["eps", "x0", "v"]
        self.parameterization = parameterization
        self.image_size = image_size
        self.channels = channels
        self.v_posterior = v_posterior
        self.loss_type = loss_type
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2,
                          cosine_s=8e-3):
        """ddpm.py:138-192 (f64 on the host, buffers stored as f32)."""
        betas = given_betas if exists(given_betas) else make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start,
                                                                         linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.linear_start = linear_start
        self.linear_end = linear_end
        to_torch = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer('betas', to_torch(betas))
        self.register_buffer('alphas_cumprod', to_torch(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_torch(alphas_cumprod_prev))
        self.register_buffer('sqrt_alphas_cumprod', to_torch(np.sqrt(alphas_cumprod)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_torch(np.sqrt(1. - alphas_cumprod)))
        self.register_buffer('log_one_minus_alphas_cumprod', to_torch(np.log(1. - alphas_cumprod)))
        self.register_buffer('sqrt_recip_alphas_cumprod', to_torch(np.sqrt(1. / alphas_cumprod)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', to_torch(np.sqrt(1. / alphas_cumprod - 1)))

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:356-359 as one HIP kernel."""
        noise = default(noise, lambda: torch.randn_like(x_start))
        return ops.q_sample(x_start.float(), noise.float(), self.sqrt_alphas_cumprod.gather(-1, t),
                            self.sqrt_one_minus_alphas_cumprod.gather(-1, t))

    def get_loss(self, pred, target, mean=True):
        """ddpm.py:367-380 (l2, mean) -> fp32 scalar on device."""
        if self.loss_type != 'l2' or not mean:
            raise NotImplementedError("only the mean l2 (eps-MSE) loss is on the AnyEdit training path (train.py:696)")
        return ops.mse(pred, target)


class LatentDiffusion(DDPM):
    """ddpm.py:522-932 reduced to what the samplers and the training step call."""

    def __init__(self, unet_config, conditioning_key="crossattn", scale_factor=1.0, first_stage_config=None,
                 cond_stage_config=None, **kwargs):
        super().__init__(unet_config, conditioning_key=conditioning_key, **{k: v for k, v in kwargs.items()
                                                                         if k not in ("force_null_conditioning", "use_ema")})
        self.scale_factor = scale_factor
        self.first_stage_model = None
        self.cond_stage_model = None   # CLIP text tower: outside the hot path (the caller passes its hidden states)
        if first_stage_config is not None:
            self.instantiate_first_stage(first_stage_config)

    def instantiate_first_stage(self, config):
        """ddpm.py:615-620.  `config` is an {'target', 'params'} dict resolved inside this package (ldm.* -> anyedit_amd.ldm.*)
        or an already built first-stage module."""
        from anyedit_amd.ldm.util import instantiate_from_config
        model = config if isinstance(config, nn.Module) else instantiate_from_config(config)
        self.first_stage_model = model.eval()
        for param in self.first_stage_model.parameters():
            param.requires_grad = False

    def get_first_stage_encoding(self, encoder_posterior):
        """ddpm.py:655-662."""
        from anyedit_amd.ldm.modules.distributions.distributions import DiagonalGaussianDistribution
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:822-830."""
        if predict_cids:
            raise NotImplementedError("VQ first stages are not on the AnyEdit path")
        z = 1. / self.scale_factor * z
        return self.first_stage_model.decode(z)

    @torch.no_grad()
    def encode_first_stage(self, x):
        """ddpm.py:832-834."""
        return self.first_stage_model.encode(x)

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """ddpm.py:854-869."""
        if isinstance(cond, dict):
            pass
        else:
            if not isinstance(cond, list):
                cond = [cond]
            key = 'c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn'
            cond = {key: cond}
        x_recon = self.model(x_noisy, t, **cond)
        if isinstance(x_recon, tuple) and not return_ids:
            return x_recon[0]
        return x_recon

    def p_losses(self, x_start, cond, t, noise=None):
        """ddpm.py:889-932 reduced to loss_simple (eps target), what train.py:694-696 computes."""
        noise = default(noise, lambda: torch.randn_like(x_start))
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        model_output = self.apply_model(x_noisy, t, cond)
        if self.parameterization != "eps":
            raise NotImplementedError()
        loss = self.get_loss(model_output, noise, mean=True)
        return loss, {"loss_simple": loss}
