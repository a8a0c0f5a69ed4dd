"""Oracle (test infrastructure): DDIM sampler loop, q_sample, eps-MSE — fp32 CPU restatement.

Follows ldm/models/diffusion/ddim.py:54-251 (sample / ddim_sampling / p_sample_ddim),
:300-336 (stochastic_encode / decode), ldm/models/diffusion/ddpm.py:356-359 (q_sample),
:367-380 + :889-932 (eps-MSE), the 3-branch InstructPix2Pix combine of
AnyEdit_Collection/adaptive_editing_pipelines/tools/global_tool.py:166-184, and the
conditioning-dropout masks of train.py:652-669.

RNG contract (G11): x_T = randn(shape) if not given; per step: [mask path] randn_like(x0) inside
q_sample, then the UNet, then randn(shape) for the sigma term EVEN when sigma == 0.
"""
import numpy as np
import torch

from . import schedule_ref as S


def extract(a, t, x_shape):
    """util.py:96-99."""
    out = a.gather(-1, t)
    return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def q_sample(buffers, x_start, t, noise=None):
    """ddpm.py:356-359."""
    if noise is None:
        noise = torch.randn_like(x_start)
    return (extract(buffers["sqrt_alphas_cumprod"], t, x_start.shape) * x_start +
            extract(buffers["sqrt_one_minus_alphas_cumprod"], t, x_start.shape) * noise)


def _cat_cond(uc, c):
    """ddim.py:193-210: unconditional FIRST, conditional second."""
    if isinstance(c, dict):
        out = {}
        for k in c:
            if isinstance(c[k], list):
                out[k] = [torch.cat([uc[k][i], c[k][i]]) for i in range(len(c[k]))]
            else:
                out[k] = torch.cat([uc[k], c[k]])
        return out
    if isinstance(c, list):
        return [torch.cat([uc[i], c[i]]) for i in range(len(c))]
    return torch.cat([uc, c])


def get_v(buffers, x, noise, t):
    """ddpm.py:361-365: the v-prediction target sqrt(acp_t) noise - sqrt(1 - acp_t) x."""
    return extract(buffers["sqrt_alphas_cumprod"], t, x.shape) * noise - extract(buffers["sqrt_one_minus_alphas_cumprod"], t, x.shape) * x


def predict_start_from_z_and_v(buffers, x_t, t, v):
    """ddpm.py:290-296."""
    return extract(buffers["sqrt_alphas_cumprod"], t, x_t.shape) * x_t - extract(buffers["sqrt_one_minus_alphas_cumprod"], t, x_t.shape) * v


def predict_eps_from_z_and_v(buffers, x_t, t, v):
    """ddpm.py:298-302."""
    return extract(buffers["sqrt_alphas_cumprod"], t, x_t.shape) * v + extract(buffers["sqrt_one_minus_alphas_cumprod"], t, x_t.shape) * x_t


def p_sample_ddim(apply_model, sched, x, c, t, index, scale=1.0, uc=None, temperature=1.0, v_buffers=None):
    """ddim.py:180-251.  `v_buffers`: the model's schedule buffers when the network predicts v (parameterization == "v", :214-217,
    :232-235): eps and x0 then come from the model's own sqrt(acp_t) tables at the network timestep, not from the DDIM tables."""
    b = x.shape[0]
    if uc is None or scale == 1.0:
        e_t = apply_model(x, t, c)
    else:
        x_in = torch.cat([x] * 2)
        t_in = torch.cat([t] * 2)
        e_uncond, e_cond = apply_model(x_in, t_in, _cat_cond(uc, c)).chunk(2)
        e_t = e_uncond + scale * (e_cond - e_uncond)
    model_output = e_t
    if v_buffers is not None:
        e_t = predict_eps_from_z_and_v(v_buffers, x, t, model_output)
    full = lambda v: torch.full((b, 1, 1, 1), v)  # torch.full casts python/np floats to float32 (G5)
    a_t = full(sched["ddim_alphas"][index])
    a_prev = full(sched["ddim_alphas_prev"][index])
    sigma_t = full(sched["ddim_sigmas"][index])
    sqrt_one_minus_at = full(sched["ddim_sqrt_one_minus_alphas"][index])
    if v_buffers is None:
        pred_x0 = (x - sqrt_one_minus_at * e_t) / a_t.sqrt()
    else:
        pred_x0 = predict_start_from_z_and_v(v_buffers, x, t, model_output)
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    noise = sigma_t * torch.randn(x.shape) * temperature
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + noise
    return x_prev, pred_x0


def ddim_sample(apply_model, buffers, S_steps, shape, cond, eta=0.0, x_T=None, scale=1.0, uc=None,
                mask=None, x0=None, log_every_t=100, temperature=1.0, parameterization="eps"):
    """ddim.py:54-178.  Returns (img, intermediates, schedule)."""
    sched = S.make_ddim_schedule(buffers, S_steps, "uniform", eta)
    b = shape[0]
    img = torch.randn(shape) if x_T is None else x_T
    timesteps = sched["ddim_timesteps"]
    inter = {"x_inter": [img], "pred_x0": [img]}
    time_range = np.flip(timesteps)
    total_steps = timesteps.shape[0]
    for i, step in enumerate(time_range):
        index = total_steps - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            img_orig = q_sample(buffers, x0, ts)
            img = img_orig * mask + (1.0 - mask) * img
        img, pred_x0 = p_sample_ddim(apply_model, sched, img, cond, ts, index, scale, uc, temperature,
                                     v_buffers=buffers if parameterization == "v" else None)
        if index % log_every_t == 0 or index == total_steps - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter, sched


def stochastic_encode(sched, x0, t, noise):
    """ddim.py:300-314 (use_original_steps=False)."""
    sa = torch.sqrt(sched["ddim_alphas"])
    s1 = torch.as_tensor(sched["ddim_sqrt_one_minus_alphas"])
    return extract(sa, t, x0.shape) * x0 + extract(s1, t, x0.shape) * noise


def ddim_encode(apply_model, sched, buffers, x0, c, t_enc, use_original_steps=False, return_intermediates=None, scale=1.0, uc=None,
                timestep_from_schedule=False):
    """ddim.py:253-298 (DDIM inversion).  Kept quirks: the network is queried at t = LOOP INDEX i, not at ddim_timesteps[i];
    alphas_next is the fp32 schedule tensor, alphas the float64 array of previous alphas (G5 dtype mix), so the two
    coefficients are formed in float64 and rounded to fp32 when they meet the fp32 latents."""
    num_reference_steps = buffers["alphas_cumprod"].shape[0] if use_original_steps else sched["ddim_timesteps"].shape[0]
    assert t_enc <= num_reference_steps
    num_steps = t_enc
    if use_original_steps:
        alphas_next = buffers["alphas_cumprod"][:num_steps]
        alphas = buffers["alphas_cumprod_prev"][:num_steps]
    else:
        alphas_next = torch.as_tensor(sched["ddim_alphas"])[:num_steps]
        alphas = torch.tensor(np.asarray(sched["ddim_alphas_prev"])[:num_steps])
    x_next = x0
    intermediates, inter_steps = [], []
    for i in range(num_steps):
        # cldm/ddim_hacked.py:237-254 (the AnyDoor sampler) queries the network at the schedule's timestep instead of the loop index
        ti = i if not timestep_from_schedule else (i if use_original_steps else int(sched["ddim_timesteps"][i]))
        t = torch.full((x0.shape[0],), ti, dtype=torch.long)
        if scale == 1.0:
            noise_pred = apply_model(x_next, t, c)
        else:
            e_t_uncond, noise_pred = torch.chunk(apply_model(torch.cat((x_next, x_next)), torch.cat((t, t)), _cat_cond(uc, c)), 2)
            noise_pred = e_t_uncond + scale * (noise_pred - e_t_uncond)
        xt_weighted = (alphas_next[i] / alphas[i]).sqrt() * x_next
        weighted_noise_pred = alphas_next[i].sqrt() * ((1 / alphas_next[i] - 1).sqrt() - (1 / alphas[i] - 1).sqrt()) * noise_pred
        x_next = xt_weighted + weighted_noise_pred
        if return_intermediates and i % (num_steps // return_intermediates) == 0 and i < num_steps - 1:
            intermediates.append(x_next)
            inter_steps.append(i)
        elif return_intermediates and i >= num_steps - 2:
            intermediates.append(x_next)
            inter_steps.append(i)
    out = {"x_encoded": x_next, "intermediate_steps": inter_steps}
    if return_intermediates:
        out["intermediates"] = intermediates
    return x_next, out


def ddim_decode(apply_model, sched, x_latent, cond, t_start, scale=1.0, uc=None):
    """ddim.py:316-336."""
    timesteps = sched["ddim_timesteps"][:t_start]
    time_range = np.flip(timesteps)
    total = timesteps.shape[0]
    x = x_latent
    for i, step in enumerate(time_range):
        index = total - i - 1
        ts = torch.full((x.shape[0],), int(step), dtype=torch.long)
        x, _ = p_sample_ddim(apply_model, sched, x, cond, ts, index, scale, uc)
    return x


# ------------------------------------------------------------------ IP2P / AnySD 3-branch loop
def ip2p_combine(e_text, e_image, e_uncond, s_txt, s_img):
    """global_tool.py:172-177."""
    return e_uncond + s_txt * (e_text - e_image) + s_img * (e_image - e_uncond)


def ip2p_edit_loop(unet_fn, buffers, S_steps, x_T, img_lat, ctx, null_ctx, s_txt=7.5, s_img=1.5,
                   eta=0.0, mask=None, x0=None, noise_for_blend=None):
    """3-branch CFG edit loop: batch order [text+image, image-only, uncond] (global_tool.py:166-184,
    290-304) driven by the ldm DDIM update (ddim.py:223-250) — SURVEY.md Appendix B recipe.

    unet_fn(x[3B,8,h,w], t[3B], context[3B,L,D]) -> eps[3B,4,h,w].
    Masked-latent blend per step (global_tool.py:183-184): latents*mask + q_sample(x0, t_i; noise)*(1-mask).
    """
    sched = S.make_ddim_schedule(buffers, S_steps, "uniform", eta)
    B = x_T.shape[0]
    ts_arr = sched["ddim_timesteps"]
    total = ts_arr.shape[0]
    img = x_T
    text_embedding = torch.cat([ctx, null_ctx, null_ctx], dim=0)
    img_cond = torch.cat([img_lat, img_lat, torch.zeros_like(img_lat)], dim=0)
    for i, step in enumerate(np.flip(ts_arr)):
        index = total - i - 1
        t = torch.full((3 * B,), int(step), dtype=torch.long)
        x_in = torch.cat([torch.cat([img] * 3), img_cond], dim=1)
        e_t, e_i, e_u = unet_fn(x_in, t, text_embedding).chunk(3)
        e = ip2p_combine(e_t, e_i, e_u, s_txt, s_img)
        a_t = torch.full((B, 1, 1, 1), sched["ddim_alphas"][index])
        a_prev = torch.full((B, 1, 1, 1), sched["ddim_alphas_prev"][index])
        sigma_t = torch.full((B, 1, 1, 1), sched["ddim_sigmas"][index])
        s1 = torch.full((B, 1, 1, 1), sched["ddim_sqrt_one_minus_alphas"][index])
        pred_x0 = (img - s1 * e) / a_t.sqrt()
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e
        noise = sigma_t * torch.randn(img.shape)
        img = a_prev.sqrt() * pred_x0 + dir_xt + noise
        if mask is not None:
            tt = torch.full((B,), int(step), dtype=torch.long)
            tmp = q_sample(buffers, x0, tt, noise=noise_for_blend)
            img = img * mask + tmp * (1.0 - mask)
    return img


# ------------------------------------------------------------------ A11 training-step arithmetic
def eps_mse(pred, target):
    """train.py:696 / ddpm.py:367-380: mean((pred-target)^2) in fp32."""
    return torch.mean((pred.float() - target.float()) ** 2)


def conditioning_dropout_masks(random_p, p):
    """train.py:652-669.  random_p: one U(0,1) draw per sample.
    prompt_mask (True -> replace text cond by null): random_p < 2p
    image_mask  (multiplies the original-image latents): 1 - [p <= random_p < 3p]."""
    prompt_mask = random_p < 2 * p
    image_mask = 1 - ((random_p >= p).float() * (random_p < 3 * p).float())
    return prompt_mask, image_mask
