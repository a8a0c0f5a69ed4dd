"""Oracle (test infrastructure): diffusion schedules and DDIM integer bookkeeping, numpy/f64.

Follows ldm/modules/diffusionmodules/util.py:21-74 and ldm/models/diffusion/ddpm.py:138-192,
ldm/models/diffusion/ddim.py:23-52.  Integer arrays must be bit-exact with the reference.
"""
import math

import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """util.py:21-43.  The reference computes in torch.float64; linspace in numpy f64 is identical to
    the last ulp only if built the same way, so we use torch here too (plumbing, CPU)."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "cosine":
        timesteps = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = timesteps / (1 + cosine_s) * np.pi / 2
        alphas = torch.cos(alphas).pow(2)
        alphas = alphas / alphas[0]
        betas = 1 - alphas[1:] / alphas[:-1]
        betas = torch.clamp(betas, min=0, max=0.999)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps):
    """util.py:46-60.  NOTE (G4): length != S when S does not divide 1000 (the assert is commented out)."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(ddim_discr_method)
    return ddim_timesteps + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:63-74 (alphacums: numpy f64 or torch f32 tensor, as the caller passes)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def register_schedule(beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """ddpm.py:138-169: the float32 buffers a DDPM model exposes (subset used on the hot path)."""
    betas = make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s)
    alphas = 1.0 - betas
    alphas_cumprod = np.cumprod(alphas, axis=0)
    alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(alphas_cumprod),
        "alphas_cumprod_prev": f32(alphas_cumprod_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(alphas_cumprod)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - alphas_cumprod)),
        "num_timesteps": int(betas.shape[0]),
    }


def make_ddim_schedule(model_buffers, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0):
    """ddim.py:23-52.  G5: ddim_alphas = float32 torch tensor (indexing a f32 tensor with a numpy int
    array), ddim_alphas_prev = float64 numpy, ddim_sigmas = float64 torch tensor."""
    ts = make_ddim_timesteps(ddim_discretize, ddim_num_steps, model_buffers["num_timesteps"])
    ac = model_buffers["alphas_cumprod"].cpu()
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, ddim_eta)
    return {
        "ddim_timesteps": ts,
        "ddim_sigmas": sigmas,
        "ddim_alphas": alphas,
        "ddim_alphas_prev": alphas_prev,
        "ddim_sqrt_one_minus_alphas": np.sqrt(1.0 - alphas),
    }


def timestep_embedding(timesteps, dim, max_period=10000):
    """util.py:154-174; order is [cos, sin] (G7)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb
