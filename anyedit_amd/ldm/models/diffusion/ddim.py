"""Mirror of ldm/models/diffusion/ddim.py (DDIMSampler) — host loop + fused HIP update kernel.

Integer bookkeeping (timestep table, index = total - i - 1, ts = full((B,), step)) is the reference's, bit for bit
(ddim.py:23-52, 150-152).  Per step the reference issues ~15 small torch ops (4x torch.full, chunk, CFG combine, sqrt,
mul/add chains, ddim.py:211-250); here that is ONE kernel (ae_ddim_step_f32) whose fp32 arithmetic is unfused so it
matches the reference expression order exactly, with the schedule coefficients rounded to fp32 on the host exactly
where `torch.full` rounds them (G5).  RNG consumption order follows G11 (randn drawn even when sigma == 0).
"""
import numpy as np
import torch

from anyedit_amd import ops
from anyedit_amd.ldm.util import warn_conditioning_batch
from anyedit_amd.ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps


def _f32(v):
    """What torch.full((b,1,1,1), v) stores: the value rounded to float32."""
    if isinstance(v, torch.Tensor):
        return np.float32(v.detach().cpu().to(torch.float32).item())
    return np.float32(v)


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.randn = torch.randn  # tests may replace this to replay a CPU noise stream

    def register_buffer(self, name, attr):
        # ddim.py:17-21 force-moves tensors to "cuda"; here: to the model's device
        if type(attr) == torch.Tensor:
            dev = self.model.device if hasattr(self.model, "device") else attr.device
            if attr.device != torch.device(dev):
                attr = attr.to(dev)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """ddim.py:23-52."""
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        to_torch = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        self.register_buffer('betas', to_torch(self.model.betas))
        self.register_buffer('alphas_cumprod', to_torch(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_torch(self.model.alphas_cumprod_prev))
        ac = alphas_cumprod.cpu()
        self.register_buffer('sqrt_alphas_cumprod', to_torch(np.sqrt(ac)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_torch(np.sqrt(1. - ac)))
        self.register_buffer('log_one_minus_alphas_cumprod', to_torch(np.log(1. - ac)))
        self.register_buffer('sqrt_recip_alphas_cumprod', to_torch(np.sqrt(1. / ac)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', to_torch(np.sqrt(1. / ac - 1)))
        ddim_sigmas, ddim_alphas, ddim_alphas_prev = make_ddim_sampling_parameters(alphacums=ac, ddim_timesteps=self.ddim_timesteps,
                                                                                   eta=ddim_eta, verbose=verbose)
        # host copies: the per-step coefficients are scalars consumed by the update kernel (no device round trip)
        self.ddim_sigmas = ddim_sigmas
        self.ddim_alphas = ddim_alphas
        self.ddim_alphas_prev = ddim_alphas_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - ddim_alphas)
        acp = self.model.alphas_cumprod_prev.cpu().to(torch.float32)
        acf = ac.to(torch.float32)
        self.ddim_sigmas_for_original_num_steps = ddim_eta * torch.sqrt((1 - acp) / (1 - acf) * (1 - acf / acp))

    # ------------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, **kwargs):
        """ddim.py:54-120."""
        warn_conditioning_batch(conditioning, batch_size)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  dynamic_threshold=dynamic_threshold, ucg_schedule=ucg_schedule)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None):
        """ddim.py:122-178."""
        device = self.model.betas.device
        b = shape[0]
        img = self.randn(shape, device=device) if x_T is None else x_T
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif timesteps is not None and not ddim_use_original_steps:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = reversed(range(0, timesteps)) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                # img_orig = q_sample(x0, ts); img = img_orig*mask + (1-mask)*img   (ddim.py:154-157), one kernel
                noise = self.randn(x0.shape, device=device)
                sa = float(_f32(self.model.sqrt_alphas_cumprod[int(step)]))
                s1 = float(_f32(self.model.sqrt_one_minus_alphas_cumprod[int(step)]))
                img = ops.mask_blend(img, x0, noise, mask, sa, s1)   # any mask that broadcasts against img ([B,1,H,W] or [B,C,H,W])
            if ucg_schedule is not None:
                assert len(ucg_schedule) == len(time_range)
                unconditional_guidance_scale = ucg_schedule[i]
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                              quantize_denoised=quantize_denoised, temperature=temperature,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning,
                                              dynamic_threshold=dynamic_threshold)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    def _coeffs(self, index, use_original_steps):
        """fp32 scalars exactly as ddim.py:228-250 forms them (torch.full -> fp32, then fp32 tensor arithmetic)."""
        if use_original_steps:
            a_t = _f32(self.model.alphas_cumprod[index])
            a_prev = _f32(self.model.alphas_cumprod_prev[index])
            s1m = _f32(self.model.sqrt_one_minus_alphas_cumprod[index])
            sigma = _f32(self.ddim_sigmas_for_original_num_steps[index])
        else:
            a_t = _f32(self.ddim_alphas[index])
            a_prev = _f32(self.ddim_alphas_prev[index])
            s1m = _f32(self.ddim_sqrt_one_minus_alphas[index])
            sigma = _f32(self.ddim_sigmas[index])
        sqrt_at = np.sqrt(a_t, dtype=np.float32)
        sqrt_a_prev = np.sqrt(a_prev, dtype=np.float32)
        one = np.float32(1.0)
        dir_coef = np.sqrt(one - a_prev - sigma * sigma, dtype=np.float32)  # (1. - a_prev - sigma_t**2).sqrt()
        return float(s1m), float(sqrt_at), float(sqrt_a_prev), float(dir_coef), float(sigma)

    def _model_eps(self, x, c, t, unconditional_guidance_scale, unconditional_conditioning):
        """ddim.py:187-212: one network call on the [uncond, cond] batch.  Returns (eps [branches*B, ...], branches)."""
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(x, t, c), 1
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t] * 2)
        if isinstance(c, dict):
            assert isinstance(unconditional_conditioning, dict)
            c_in = dict()
            for k in c:
                if isinstance(c[k], list):
                    c_in[k] = [torch.cat([unconditional_conditioning[k][i], c[k][i]]) for i in range(len(c[k]))]
                else:
                    c_in[k] = torch.cat([unconditional_conditioning[k], c[k]])
        elif isinstance(c, list):
            assert isinstance(unconditional_conditioning, list)
            c_in = [torch.cat([unconditional_conditioning[i], c[i]]) for i in range(len(c))]
        else:
            c_in = torch.cat([unconditional_conditioning, c])
        return self.model.apply_model(x_in, t_in, c_in), 2  # [2B,...] = [uncond, cond]

    def _encode_eps(self, x, t, c, unconditional_conditioning):
        """ddim.py:276-280: tensor conditionings only, batch order [uncond, cond]."""
        return self.model.apply_model(torch.cat((x, x)), torch.cat((t, t)), torch.cat((unconditional_conditioning, c)))

    def _encode_timestep(self, i, use_original_steps):
        """ddim.py:272: the inversion loop hands the LOOP INDEX to the network as the timestep (reference quirk, kept)."""
        return i

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None):
        """ddim.py:180-251."""
        b, device = x.shape[0], x.device
        if quantize_denoised:
            raise NotImplementedError("quantize_denoised needs a VQ first stage (ddim.py:234-235); the AnyEdit path decodes with AutoencoderKL")
        if dynamic_threshold is not None:
            raise NotImplementedError()
        param = getattr(self.model, "parameterization", "eps")
        if param not in ("eps", "v"):
            raise NotImplementedError("x0-parameterisation: DDIMSampler handles eps- and v-prediction models (ddim.py:214-217)")
        eps, branches = self._model_eps(x, c, t, unconditional_guidance_scale, unconditional_conditioning)
        noise = self.randn((1, *x.shape[1:]), device=device).repeat(b, 1, 1, 1) if repeat_noise else self.randn(x.shape, device=device)
        if noise_dropout > 0.:
            # ddim.py:246-247 drops (and rescales) sigma_t * noise * temperature; the mask and its 1 / (1 - p) commute with those two scalars, which the
            # fused step kernel applies
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        coeffs = self._coeffs(index, use_original_steps)
        if score_corrector is not None:
            # ddim.py:219-221: the corrector sees the guidance-combined eps — one launch writes it out, the corrector runs, the update takes ONE branch
            assert param == "eps", 'not implemented'
            _, _, e_t = ops.ddim_step(x.float(), eps.float(), coeffs, branches, s0=float(unconditional_guidance_scale), noise=noise.float().contiguous(),
                                      temperature=float(temperature), want_pred_x0=False, want_e=True)
            e_t = score_corrector.modify_score(self.model, e_t, x, t, c, **(corrector_kwargs or {}))
            return ops.ddim_step(x.float(), e_t.float().contiguous(), coeffs, 1, noise=noise.float().contiguous(), temperature=float(temperature))
        if param == "v":
            return self._v_step(x.float().contiguous(), eps.float(), t, coeffs, branches, float(unconditional_guidance_scale),
                                noise.float().contiguous(), float(temperature))
        x_prev, pred_x0 = ops.ddim_step(x.float(), eps.float(), coeffs, branches, s0=float(unconditional_guidance_scale),
                                        noise=noise.float().contiguous(), temperature=float(temperature))
        return x_prev, pred_x0

    def _v_step(self, x, out, t, coeffs, branches, scale, noise, temperature):
        """ddim.py:212-217, 232-250 for a v-prediction network: guided v = v_u + s (v_c - v_u); eps and x0 from the MODEL's sqrt(acp_t)
        tables at the network timestep (predict_eps_from_z_and_v / predict_start_from_z_and_v), then the usual DDIM combination."""
        if branches == 2:
            vu, vc = out[:x.shape[0]].contiguous(), out[x.shape[0]:].contiguous()
            if vc.data_ptr() % 16:   # the conditional half is a slice: 16-byte aligned only when B*C*H*W % 4 == 0 (ops.lincomb takes 16-byte pointers)
                vc = vc.clone()
            v = ops.lincomb([(vu, 1.0), (ops.lincomb([(vc, 1.0), (vu, -1.0)]), scale)])
        else:
            v = out.contiguous()
        e_t = self.model.predict_eps_from_z_and_v(x, t, v)
        pred_x0 = self.model.predict_start_from_z_and_v(x, t, v)
        _, _, sqrt_a_prev, dir_coef, sigma = coeffs
        x_prev = ops.lincomb([(pred_x0, sqrt_a_prev), (e_t, dir_coef), (noise, sigma * temperature)])
        return x_prev, pred_x0

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:300-314."""
        if use_original_steps:
            sa = self.sqrt_alphas_cumprod
            s1 = self.sqrt_one_minus_alphas_cumprod
        else:
            sa = torch.sqrt(torch.as_tensor(self.ddim_alphas).float()).to(x0.device)
            s1 = torch.as_tensor(self.ddim_sqrt_one_minus_alphas).float().to(x0.device)
        if noise is None:
            noise = self.randn(x0.shape, device=x0.device)
        return ops.q_sample(x0.float(), noise.float(), sa.gather(-1, t).float(), s1.gather(-1, t).float())

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """ddim.py:316-336."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        x_dec = x_latent
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec

    @torch.no_grad()
    def encode(self, x0, c, t_enc, use_original_steps=False, return_intermediates=None, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, callback=None):
        """ddim.py:253-298 (DDIM inversion).  Host loop over apply_model; the update x_next = cx*x + ce*e is one fused kernel
        per step (ae_ddim_encode_step_f32).  Reference quirks kept: t = loop index i (not ddim_timesteps[i]); cx, ce formed in
        float64 from the fp32 alphas_next and the float64 alphas_prev, rounded to fp32 where they meet the latents."""
        num_reference_steps = self.ddpm_num_timesteps if use_original_steps else self.ddim_timesteps.shape[0]
        assert t_enc <= num_reference_steps
        num_steps = t_enc
        if use_original_steps:
            a_next = np.asarray(self.alphas_cumprod[:num_steps].detach().cpu(), dtype=np.float32)
            a_prev = np.asarray(self.alphas_cumprod_prev[:num_steps].detach().cpu(), dtype=np.float32)  # fp32 buffer in this branch
        else:
            a_next = np.asarray(self.ddim_alphas[:num_steps], dtype=np.float32)
            a_prev = np.asarray(self.ddim_alphas_prev[:num_steps], dtype=np.float64)
        x_next = x0.float().contiguous()
        intermediates, inter_steps = [], []
        cfg = unconditional_guidance_scale != 1.
        for i in range(num_steps):
            t = torch.full((x0.shape[0],), int(self._encode_timestep(i, use_original_steps)), device=x0.device, dtype=torch.long)
            if not cfg:
                eps = self.model.apply_model(x_next, t, c)
            else:
                assert unconditional_conditioning is not None
                eps = self._encode_eps(x_next, t, c, unconditional_conditioning)
            an32, ap = np.float32(a_next[i]), a_prev[i]
            cx = np.float32(np.sqrt(np.float64(an32) / np.float64(ap)) if not use_original_steps else np.sqrt(an32 / np.float32(ap), dtype=np.float32))
            s_next = np.sqrt(an32, dtype=np.float32)                                   # alphas_next[i].sqrt()          (fp32)
            r_next = np.sqrt(np.float32(1) / an32 - np.float32(1), dtype=np.float32)   # (1 / alphas_next[i] - 1).sqrt() (fp32)
            if use_original_steps:
                r_prev = np.sqrt(np.float32(1) / np.float32(ap) - np.float32(1), dtype=np.float32)
                ce = np.float32(s_next * (r_next - r_prev))
            else:
                r_prev = np.sqrt(1.0 / np.float64(ap) - 1.0)                            # float64
                ce = np.float32(np.float64(s_next) * (np.float64(r_next) - r_prev))
            x_next = ops.ddim_encode_step(x_next, eps.float(), float(cx), float(ce), branches=2 if cfg else 1,
                                          scale=float(unconditional_guidance_scale))
            if return_intermediates and i % (num_steps // return_intermediates) == 0 and i < num_steps - 1:
                intermediates.append(x_next)
                inter_steps.append(i)
            elif return_intermediates and i >= num_steps - 2:
                intermediates.append(x_next)
                inter_steps.append(i)
            if callback:
                callback(i)
        out = {'x_encoded': x_next, 'intermediate_steps': inter_steps}
        if return_intermediates:
            out.update({'intermediates': intermediates})
        return x_next, out
