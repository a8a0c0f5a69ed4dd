"""Oracle (test infrastructure) for SURVEY.md §8(f) N4: PLMSSampler (ldm/models/diffusion/plms.py:118-244), CPU fp32 restatement
over the oracle's DDIM schedule.  Pinned to tests/golden/plms.npz (the reference sampler run on an analytic eps model)."""
import numpy as np
import torch

from . import schedule_ref as S
from .ddim_ref import q_sample, _cat_cond


def plms_sample(apply_model, buffers, S_steps, shape, cond, x_T, scale=1.0, uc=None, mask=None, x0=None, log_every_t=100, score_corrector=None,
                noise_dropout=0.0):
    """score_corrector: callable (e_t, x, t, c) -> e_t applied to the guidance-combined eps of every network evaluation (plms.py:195-197); noise_dropout:
    plms.py:222-224 (sigma_t = 0 in PLMS: only the RNG consumption shows)."""
    sched = S.make_ddim_schedule(buffers, S_steps, "uniform", 0.0)
    timesteps = sched["ddim_timesteps"]
    time_range = np.flip(timesteps)
    total = timesteps.shape[0]
    b = shape[0]
    img = x_T
    inter = {"x_inter": [img], "pred_x0": [img]}
    old_eps = []

    def model_output(x, t):
        if uc is None or scale == 1.0:
            return apply_model(x, t, cond)
        e_u, e_c = apply_model(torch.cat([x] * 2), torch.cat([t] * 2), _cat_cond(uc, cond)).chunk(2)
        return e_u + scale * (e_c - e_u)

    if score_corrector is not None:
        raw_output = model_output

        def model_output(x, t):   # noqa: F811
            return score_corrector(raw_output(x, t), x, t, cond)

    for i, step in enumerate(time_range):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        if mask is not None:
            img = q_sample(buffers, x0, ts) * mask + (1.0 - mask) * img

        def step_from(e, x=img, index=index):
            a_t = torch.full((b, 1, 1, 1), float(sched["ddim_alphas"][index]))
            a_prev = torch.full((b, 1, 1, 1), float(sched["ddim_alphas_prev"][index]))
            sigma_t = torch.full((b, 1, 1, 1), float(sched["ddim_sigmas"][index]))
            s1m = torch.full((b, 1, 1, 1), float(sched["ddim_sqrt_one_minus_alphas"][index]))
            pred_x0 = (x - s1m * e) / a_t.sqrt()
            dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e
            noise = sigma_t * torch.randn(x.shape)
            if noise_dropout > 0.0:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            return a_prev.sqrt() * pred_x0 + dir_xt + noise, pred_x0

        e_t = model_output(img, ts)
        if len(old_eps) == 0:
            x_prov, _ = step_from(e_t)
            e_prime = (e_t + model_output(x_prov, ts_next)) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = step_from(e_prime)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter, sched
