"""Host logic of the samplers on CPU — schedule tables, step bookkeeping, history handling, coefficient arithmetic, RNG consumption order,
guidance batching — against the outputs of the reference's own samplers (tests/golden/*.npz).

The product samplers call elementwise HIP kernels for the per-step arithmetic (there is no CPU path in the product, by design).  Here — in
tests/ only — those kernels are replaced by `_StandIns`: the formulas their headers in csrc/elementwise.hip state, written in torch, so that
everything AROUND the kernels (the part that is Python in the product as well) runs on a box without a GPU.  Whether the kernels compute
those formulas is what the `-m gpu` twins of these tests check (tests/test_hip_unet.py::test_plms_sampler_golden, …_ddim_sampler_v_prediction_golden,
…_ddim_hacked_sampler_golden, …_dpm_solver_sampler_golden, …_dpm_solver_general_variants_golden)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, T
from dpm_cases import DPM_GENERAL_CASES
from oracle import schedule_ref as S


class _StandIns:
    """torch statements of the sampler kernels' documented arithmetic (csrc/elementwise.hip:96-222, 418-470)."""

    @staticmethod
    def _guided(eps, n_like, branches, s0, s1=0.0):
        parts = eps.reshape(branches, *n_like.shape)
        if branches == 1:
            return parts[0]
        if branches == 2:                                                   # [uncond, cond]
            return parts[0] + s0 * (parts[1] - parts[0])
        et, ei, eu = parts                                                   # [text, image, uncond]
        return eu + s0 * (et - ei) + s1 * (ei - eu)

    @classmethod
    def ddim_step(cls, x, eps, coeffs, branches, s0=1.0, s1=0.0, noise=None, temperature=1.0, want_pred_x0=True, want_e=False):
        s1m, sat, sap, dirc, sig = (torch.tensor(c, dtype=torch.float32) for c in coeffs)
        e = cls._guided(eps, x, branches, s0, s1)
        px0 = (x - s1m * e) / sat
        nz = sig * (noise if noise is not None else torch.zeros_like(x)) * temperature
        x_prev = sap * px0 + dirc * e + nz
        return (x_prev, px0, e) if want_e else (x_prev, px0)

    @classmethod
    def ddim_encode_step(cls, x, eps, cx, ce, branches=1, scale=1.0):
        e = cls._guided(eps, x, branches, scale)
        return torch.tensor(cx, dtype=torch.float32) * x + torch.tensor(ce, dtype=torch.float32) * e

    @staticmethod
    def plms_combine(e_t, old_eps):
        o = list(old_eps[::-1][:3])
        if len(o) == 1:
            return (3.0 * e_t - o[0]) / 2.0
        if len(o) == 2:
            return ((23.0 * e_t - 16.0 * o[0]) + 5.0 * o[1]) / 12.0
        return (((55.0 * e_t - 59.0 * o[0]) + 37.0 * o[1]) - 9.0 * o[2]) / 24.0

    @staticmethod
    def plms_combine_first(e_t, e_t_next):
        return (e_t + e_t_next) / 2.0

    @staticmethod
    def mask_blend(img, x0, noise, mask, sqrt_ac, sqrt_one_minus_ac, ip2p_order=False):
        q = torch.tensor(sqrt_ac, dtype=torch.float32) * x0 + torch.tensor(sqrt_one_minus_ac, dtype=torch.float32) * noise
        m = mask.float()
        return img * m + q * (1.0 - m) if ip2p_order else q * m + (1.0 - m) * img

    @staticmethod
    def q_sample(x0, noise, sqrt_ac_t, sqrt_one_minus_ac_t):
        shape = (-1,) + (1,) * (x0.dim() - 1)
        return sqrt_ac_t.reshape(shape) * x0 + sqrt_one_minus_ac_t.reshape(shape) * noise

    @staticmethod
    def dpm_multistep(x, model_out, branches, scale, sigma_s, alpha_s, predict_x0=True, v_param=False, m_prev=None, update=None, want_m=True):
        f = lambda v: torch.tensor(v, dtype=torch.float32)
        parts = model_out.reshape(branches, *x.shape)
        conv = (lambda o: f(alpha_s) * o + f(sigma_s) * x) if v_param else (lambda o: o)
        e = conv(parts[0])
        if branches == 2:
            e = e + f(scale) * (conv(parts[1]) - e)
        m = (x - f(sigma_s) * e) / f(alpha_s) if predict_x0 else e
        xn = None
        if update is not None:
            a, b, c, inv_r0 = update
            xn = f(a) * x - f(b) * m
            if m_prev is not None:
                xn = xn - f(c) * (f(inv_r0) * (m - m_prev))
        return (m if want_m else None), xn

    @staticmethod
    def lincomb(terms, out=None):
        acc = None
        for t, c in terms:
            if t is None:
                continue
            term = torch.tensor(float(c), dtype=torch.float32) * t
            acc = term if acc is None else acc + term
        return acc

    @staticmethod
    def dpm_adaptive_err(x_lower, x_higher, x_prev, atol, rtol):
        delta = torch.max(torch.full_like(x_lower, atol), rtol * torch.max(x_lower.abs(), x_prev.abs()))
        d = ((x_higher - x_lower) / delta).reshape(x_lower.shape[0], -1)
        return torch.sqrt((d * d).mean(1))


@pytest.fixture()
def standin_ops(monkeypatch):
    from anyedit_amd import ops
    for name in ("ddim_step", "ddim_encode_step", "plms_combine", "plms_combine_first", "mask_blend", "q_sample", "dpm_multistep", "lincomb",
                 "dpm_adaptive_err"):
        assert hasattr(ops, name), name
        monkeypatch.setattr(ops, name, getattr(_StandIns, name))
    return ops


def _analytic(x, t, c):
    return torch.sin(x.float() * 1.7 + t.float()[:, None, None, None] * 0.01) * 0.5 + c.float()[:, :, None, None] * x.float()


class _Model:
    parameterization = "eps"

    def __init__(self):
        self.num_timesteps = 1000
        for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
            if isinstance(v, torch.Tensor):
                setattr(self, k, v)
        self.device = torch.device("cpu")
        self.calls, self.seen_t = 0, []

    def apply_model(self, x, t, c):
        self.calls += 1
        self.seen_t.append(int(t[0]))
        return _analytic(x, t, c)


def _maxabs(a, b):
    return float((a - b).abs().max())


def test_plms_sampler_host_logic(standin_ops):
    from anyedit_amd.ldm.models.diffusion.plms import PLMSSampler
    g = load_golden("plms")
    sampler = PLMSSampler(_Model())
    for tag, steps, scale, use_mask in (("s7", 7, 1.0, False), ("s10_cfg", 10, 5.0, False), ("s6_cfg_mask", 6, 3.0, True)):
        kw = dict(mask=T(g[f"{tag}.mask"]), x0=T(g[f"{tag}.x0"])) if use_mask else {}
        torch.manual_seed(4321)
        samples, inter = sampler.sample(steps, 2, (4, 8, 8), T(g["c"]), eta=0.0, x_T=T(g["x_T"]), verbose=False, unconditional_guidance_scale=scale,
                                        unconditional_conditioning=T(g["uc"]) if scale != 1.0 else None, log_every_t=1, **kw)
        assert np.array_equal(sampler.ddim_timesteps, g[f"{tag}.ddim_timesteps"])
        assert len(inter["x_inter"]) == g[f"{tag}.x_inter"].shape[0]
        assert _maxabs(samples, T(g[f"{tag}.samples"])) <= 2e-5, tag
        assert _maxabs(torch.stack(inter["pred_x0"]), T(g[f"{tag}.pred_x0"])) <= 2e-5, tag
    torch.manual_seed(4321)   # score_corrector + noise_dropout (plms.py:195-197, 222-224) against the reference run with the same corrector
    samples, inter = sampler.sample(6, 2, (4, 8, 8), T(g["c"]), eta=0.0, x_T=T(g["x_T"]), verbose=False, unconditional_guidance_scale=3.0,
                                    unconditional_conditioning=T(g["uc"]), log_every_t=1, score_corrector=_Corrector(), corrector_kwargs={"gain": 1.1}, noise_dropout=0.3)
    assert _maxabs(samples, T(g["s6_cfg_corr.samples"])) <= 2e-5
    assert _maxabs(torch.stack(inter["pred_x0"]), T(g["s6_cfg_corr.pred_x0"])) <= 2e-5
    with pytest.raises(ValueError):
        sampler.make_schedule(5, ddim_eta=0.5, verbose=False)


class _Corrector:
    """tools/gen_golden.py::AnalyticCorrector."""

    def modify_score(self, model, e_t, x, t, c, gain=1.0):
        return e_t * gain - 0.05 * x + 0.01 * c[:, :, None, None]


def test_ddim_sampler_v_prediction_host_logic(standin_ops):
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM

    class VModel(_Model):
        parameterization = "v"
        _acp_pair = DDPM._acp_pair
        predict_start_from_z_and_v = DDPM.predict_start_from_z_and_v
        predict_eps_from_z_and_v = DDPM.predict_eps_from_z_and_v
        get_v = DDPM.get_v

    g = load_golden("ddim_v")
    model = VModel()
    sampler = DDIMSampler(model)
    for tag, steps, scale, eta in (("s6", 6, 1.0, 0.0), ("s8_cfg", 8, 5.0, 0.0), ("s5_cfg_eta1", 5, 3.0, 1.0)):
        torch.manual_seed(4323)
        samples, inter = sampler.sample(steps, 2, (4, 8, 8), T(g["c"]), eta=eta, x_T=T(g["x_T"]), verbose=False, unconditional_guidance_scale=scale,
                                        unconditional_conditioning=T(g["uc"]) if scale != 1.0 else None, log_every_t=1)
        assert np.array_equal(sampler.ddim_timesteps, g[f"{tag}.ddim_timesteps"])
        assert _maxabs(samples, T(g[f"{tag}.samples"])) <= 2e-5, tag
        assert _maxabs(torch.stack(inter["pred_x0"]), T(g[f"{tag}.pred_x0"])) <= 2e-5, tag
    assert _maxabs(model.get_v(T(g["x_T"]), T(g["noise"]), T(g["t"])), T(g["get_v"])) <= 1e-6
    model.parameterization = "x0"
    with pytest.raises(NotImplementedError):
        sampler.sample(2, 2, (4, 8, 8), T(g["c"]), x_T=T(g["x_T"]), verbose=False)


def test_ddim_hacked_sampler_host_logic(standin_ops):
    """cldm.ddim_hacked.DDIMSampler: two network calls per guided step, inversion queried at ddim_timesteps[i] (not the loop index)."""
    from anyedit_amd.cldm.ddim_hacked import DDIMSampler
    g = load_golden("ddim_hacked")
    model = _Model()
    sampler = DDIMSampler(model)
    torch.manual_seed(4322)
    samples, inter = sampler.sample(8, 2, (4, 8, 8), T(g["c"]), eta=0.0, x_T=T(g["x_T"]), verbose=False, unconditional_guidance_scale=5.0,
                                    unconditional_conditioning=T(g["uc"]), log_every_t=1)
    assert model.calls == int(g["s8_cfg.network_calls"]) and np.array_equal(sampler.ddim_timesteps, g["ddim_timesteps"])
    assert _maxabs(samples, T(g["s8_cfg.samples"])) <= 2e-5
    assert _maxabs(torch.stack(inter["pred_x0"]), T(g["s8_cfg.pred_x0"])) <= 2e-5
    model.seen_t = []
    x, out = sampler.encode(T(g["x_T"]), T(g["c"]), t_enc=6, return_intermediates=2)
    assert model.seen_t == [int(v) for v in g["ddim_timesteps"][:6]]
    assert _maxabs(x, T(g["enc.x"])) <= 2e-5 and out["intermediate_steps"] == g["enc.intermediate_steps"].tolist()


def test_dpm_solver_host_logic(standin_ops):
    """DPMSolverSampler (DPM-Solver++ 2M) and DPM_Solver's multistep variants: one network evaluation per step, schedule functions, history."""
    from anyedit_amd.ldm.models.diffusion.dpm_solver import DPMSolverSampler
    from anyedit_amd.ldm.models.diffusion.dpm_solver.dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver
    g = load_golden("dpm_solver")

    def near(got, key, rel=5e-6):
        ref = T(g[key])
        assert _maxabs(got, ref) <= rel * max(float(ref.abs().max()), 1.0), key

    model = _Model()
    sampler = DPMSolverSampler(model)
    for tag, steps, scale in (("s10", 10, 1.0), ("s12_cfg", 12, 5.0), ("s20_cfg", 20, 7.5)):
        model.calls = 0
        samples, none = sampler.sample(steps, 2, (4, 8, 8), T(g["c"]), x_T=T(g["x_T"]), verbose=False, unconditional_guidance_scale=scale,
                                       unconditional_conditioning=T(g["uc"]) if scale != 1.0 else None)
        assert none is None and model.calls == steps
        near(samples, f"{tag}.samples")
    ns = NoiseScheduleVP('discrete', alphas_cumprod=model.alphas_cumprod)
    mf = model_wrapper(lambda x, t, c: model.apply_model(x, t, c), ns, model_type="noise", guidance_type="classifier-free",
                       condition=T(g["c"]), unconditional_condition=T(g["uc"]), guidance_scale=3.0)
    for st in ("time_uniform", "logSNR", "time_quadratic"):
        assert _maxabs(DPM_Solver(mf, ns).get_time_steps(st, 1.0, 0.001, 10, "cpu"), T(g[f"ts.{st}"])) <= 1e-6
    near(DPM_Solver(mf, ns, predict_x0=False).sample(T(g["x_T"]), steps=9, skip_type="logSNR", method="multistep", order=2), "eps2m.samples")
    near(DPM_Solver(mf, ns, predict_x0=True).sample(T(g["x_T"]), steps=8, skip_type="time_quadratic", method="multistep", order=2,
                                                    solver_type="taylor", denoise_to_zero=True), "taylor.samples")
    near(DPM_Solver(mf, ns, predict_x0=True).sample(T(g["x_T"]), steps=6, skip_type="time_uniform", method="multistep", order=1,
                                                    t_start=0.8, t_end=0.05), "o1.samples")
    with pytest.raises(ValueError):
        DPM_Solver(mf, ns).sample(T(g["x_T"]), steps=6, method="no_such_method")


def test_dpm_solver_general_variants_host_logic(standin_ops):
    """Singlestep orders 1-3 with the order plan, singlestep_fixed, multistep order 3, adaptive step size (with its evaluation count)."""
    from anyedit_amd.ldm.models.diffusion.dpm_solver.dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver
    g = load_golden("dpm_solver_general")
    ac = S.register_schedule("linear", 1000, 0.00085, 0.0120)["alphas_cumprod"].float()
    ns = NoiseScheduleVP('discrete', alphas_cumprod=ac)
    mf = model_wrapper(_analytic, ns, model_type="noise", guidance_type="classifier-free", condition=T(g["c"]), unconditional_condition=T(g["uc"]),
                       guidance_scale=3.0)
    plan = DPM_Solver(mf, ns)
    for steps, order in ((10, 3), (9, 3), (11, 3), (7, 2), (6, 2), (5, 1)):
        ts, orders = plan.get_orders_and_timesteps_for_singlestep_solver(steps, order, "logSNR", 1.0, 0.001, "cpu")
        assert list(orders) == list(g[f"plan.{steps}.{order}.logSNR.orders"])
        assert _maxabs(ts, T(g[f"plan.{steps}.{order}.logSNR.ts"])) <= 1e-6
    for tag, (px0, kw) in DPM_GENERAL_CASES.items():
        solver = DPM_Solver(mf, ns, predict_x0=px0)
        out = solver.sample(T(g["x_T"]), **kw)
        ref = T(g[f"{tag}.samples"])
        assert _maxabs(out, ref) / float(ref.abs().max()) <= 2e-5, tag
        if kw["method"] == "adaptive":
            assert solver.last_nfe == int(g[f"{tag}.nfe"]), (tag, solver.last_nfe)
