"""PLMSSampler on the HIP path — mirror of ldm/models/diffusion/plms.py: pseudo linear multistep sampling
(Liu et al. 2022) over the same DDIM schedule; fewer network evaluations per image than DDIM at equal quality.

Host loop over `model.apply_model`; per step two fused kernels: `ae_plms_combine_f32` (Adams-Bashforth combination of the eps
history, plms.py:226-240) and `ae_ddim_step_f32` (x_prev / pred_x0, plms.py:204-223; also yields the guidance-combined eps that
goes into the history).  The integer bookkeeping (`index`, `ts`, `ts_next`, history length) and the RNG consumption order
(q_sample noise for masks, one noise draw per x_prev evaluation even at eta = 0) follow the reference exactly.
"""
import numpy as np
import torch

from anyedit_amd import ops
from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler, _f32


class PLMSSampler(DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')  # plms.py:27-28
        super().make_schedule(ddim_num_steps, ddim_discretize=ddim_discretize, ddim_eta=ddim_eta, verbose=verbose)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, **kwargs):
        """plms.py:58-116."""
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.plms_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, dynamic_threshold=dynamic_threshold)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, dynamic_threshold=None):
        """plms.py:118-176."""
        device = self.model.betas.device
        b = shape[0]
        img = self.randn(shape, device=device) if x_T is None else x_T
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif timesteps is not None and not ddim_use_original_steps:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = list(reversed(range(0, timesteps))) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        old_eps = []
        img = img.float().contiguous()
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                noise = self.randn(x0.shape, device=device)                         # q_sample's draw (plms.py:151)
                sa = float(_f32(self.model.sqrt_alphas_cumprod[int(step)]))
                s1 = float(_f32(self.model.sqrt_one_minus_alphas_cumprod[int(step)]))
                img = ops.mask_blend(img, x0, noise, mask, sa, s1)
            img, pred_x0, e_t = self.p_sample_plms(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                                   quantize_denoised=quantize_denoised, temperature=temperature,
                                                   noise_dropout=noise_dropout, score_corrector=score_corrector,
                                                   corrector_kwargs=corrector_kwargs,
                                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                                   unconditional_conditioning=unconditional_conditioning, old_eps=old_eps,
                                                   t_next=ts_next, dynamic_threshold=dynamic_threshold)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_plms(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, old_eps=None, t_next=None, dynamic_threshold=None):
        """plms.py:178-244."""
        if quantize_denoised:
            raise NotImplementedError("quantize_denoised needs a VQ first stage (plms.py:216-217); the AnyEdit path decodes with AutoencoderKL")
        if dynamic_threshold is not None:
            raise NotImplementedError("dynamic_threshold: norm_thresholding is for pixel-space models (plms.py:218-219); no AnyEdit caller passes it")
        if score_corrector is not None:
            assert getattr(self.model, "parameterization", "eps") == "eps"      # plms.py:196
        b, device = x.shape[0], x.device
        coeffs = self._coeffs(index, use_original_steps)
        cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)

        def model_output(xx, tt):
            """eps at (xx, tt); with guidance the stacked [uncond, cond] prediction."""
            if not cfg:
                return self.model.apply_model(xx, tt, c).float()
            x_in, t_in = torch.cat([xx] * 2), torch.cat([tt] * 2)
            c_in = torch.cat([unconditional_conditioning, c])
            return self.model.apply_model(x_in, t_in, c_in).float()

        def x_prev_and_pred_x0(e, branches, want_e=False):
            noise = self.randn((1, *x.shape[1:]), device=device).repeat(b, 1, 1, 1) if repeat_noise else self.randn(x.shape, device=device)
            if noise_dropout > 0.:
                # plms.py:222-224 drops (and rescales) sigma_t * noise * temperature; the mask and its 1 / (1 - p) commute with those two scalars, which the
                # fused step kernel applies
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            return ops.ddim_step(x, e, coeffs, branches, s0=float(unconditional_guidance_scale), noise=noise.float().contiguous(),
                                 temperature=float(temperature), want_e=want_e)

        if score_corrector is not None:
            # plms.py:195-197: the corrector sees the guidance-combined eps of EVERY network evaluation (also the provisional one of the first step), so the
            # fused "guidance + update" launch cannot serve; guidance is one launch (its update values unused), the corrector runs, the update takes ONE branch
            def corrected(xx, tt):
                raw_ = model_output(xx, tt)
                e_ = ops.ddim_step(xx, raw_, coeffs, 2, s0=float(unconditional_guidance_scale), want_e=True)[2] if cfg else raw_.contiguous()
                return score_corrector.modify_score(self.model, e_, xx, tt, c, **(corrector_kwargs or {})).float().contiguous()

            e_t = corrected(x, t)
            if len(old_eps) == 0:
                x_prov, _ = x_prev_and_pred_x0(e_t, 1)
                e_t_prime = ops.plms_combine_first(e_t, corrected(x_prov, t_next))
            else:
                e_t_prime = ops.plms_combine(e_t, old_eps)
            x_prev, pred_x0 = x_prev_and_pred_x0(e_t_prime, 1)
            return x_prev, pred_x0, e_t
        raw = model_output(x, t)
        if len(old_eps) == 0:
            # pseudo improved Euler: evaluate the network again at the provisional x_prev (plms.py:227-231)
            x_prov, _, e_t = x_prev_and_pred_x0(raw, 2 if cfg else 1, want_e=True)
            raw_next = model_output(x_prov, t_next)
            if cfg:  # combine the second evaluation's branches with the same guidance formula (update values unused)
                _, _, e_next = ops.ddim_step(x_prov, raw_next, coeffs, 2, s0=float(unconditional_guidance_scale), want_e=True)
            else:
                e_next = raw_next
            e_t_prime = ops.plms_combine_first(e_t, e_next)
        else:
            if cfg:
                _, _, e_t = ops.ddim_step(x, raw, coeffs, 2, s0=float(unconditional_guidance_scale), want_e=True)
            else:
                e_t = raw.contiguous()
            e_t_prime = ops.plms_combine(e_t, old_eps)
        x_prev, pred_x0 = x_prev_and_pred_x0(e_t_prime, 1)
        return x_prev, pred_x0, e_t
