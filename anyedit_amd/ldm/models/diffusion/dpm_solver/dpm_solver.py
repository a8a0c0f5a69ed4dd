"""DPM-Solver / DPM-Solver++ multistep sampling on the HIP path — mirror of ldm/models/diffusion/dpm_solver/dpm_solver.py for the
configurations its sampler front end reaches: the discrete-time VP noise schedule (NoiseScheduleVP :7-160),
classifier-free-guided noise / v prediction (model_wrapper :161-316) and the multistep solver of orders 1-2 in both parameterisations,
both solver types and all three step spacings (DPM_Solver :319-404, 463-514, 723-777, 855-877, 1040-1073).

Structure: every per-step quantity that does not depend on the latent (log-alpha interpolation, lambda, sigma, the update coefficients)
is a handful of fp32 scalars computed on the host with the reference's expressions; the latent-sized work of one step — guidance
combine, conversion to the data prediction, history difference and the x update — is ONE kernel launch after each network evaluation
(ae_dpm_multistep_f32), instead of the reference's ~20 elementwise torch ops.  Anything outside that set raises NotImplementedError
rather than running differently: singlestep / adaptive solvers, order 3, continuous-time schedules, classifier guidance, thresholding.
"""
import torch

from anyedit_amd import ops


def interpolate_fn(x, xp, yp):
    """:1104-1142: piecewise-linear y(x) through ascending keypoints, x [N, 1], xp / yp [1, K]; beyond the ends the outermost segment
    is extended.  Host-side fp32 (a few scalars per step)."""
    xk, yk = xp.reshape(-1), yp.reshape(-1)
    q = x.reshape(-1).to(xk.device)
    hi = torch.searchsorted(xk, q.contiguous()).clamp(1, xk.shape[0] - 1)
    lo = hi - 1
    return (yk[lo] + (q - xk[lo]) * (yk[hi] - yk[lo]) / (xk[hi] - xk[lo])).reshape(-1, 1)


class NoiseScheduleVP:
    """:7-160, 'discrete' schedule (what DPMSolverSampler builds from the model's alphas_cumprod, sampler.py:69).  Kept on the CPU:
    the solver only ever evaluates it at the step times."""

    def __init__(self, schedule='discrete', betas=None, alphas_cumprod=None, continuous_beta_0=0.1, continuous_beta_1=20.):
        if schedule not in ['discrete', 'linear', 'cosine']:
            raise ValueError("Unsupported noise schedule {}. The schedule needs to be 'discrete' or 'linear' or 'cosine'".format(schedule))
        if schedule != 'discrete':
            raise NotImplementedError("NoiseScheduleVP: only the discrete-time schedule is on the AnyEdit path (sampler.py:69)")
        self.schedule = schedule
        if betas is not None:
            log_alphas = 0.5 * torch.log(1 - betas.detach().float().cpu()).cumsum(dim=0)
        else:
            assert alphas_cumprod is not None
            log_alphas = 0.5 * torch.log(alphas_cumprod.detach().float().cpu())
        self.total_N = len(log_alphas)
        self.T = 1.
        self.t_array = torch.linspace(0., 1., self.total_N + 1)[1:].reshape((1, -1))
        self.log_alpha_array = log_alphas.reshape((1, -1,))

    def marginal_log_mean_coeff(self, t):
        return interpolate_fn(t.detach().float().cpu().reshape((-1, 1)), self.t_array, self.log_alpha_array).reshape((-1))

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        log_mean_coeff = self.marginal_log_mean_coeff(t)
        return log_mean_coeff - 0.5 * torch.log(1. - torch.exp(2. * log_mean_coeff))

    def inverse_lambda(self, lamb):
        lamb = lamb.detach().float().cpu()
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return interpolate_fn(log_alpha.reshape((-1, 1)), torch.flip(self.log_alpha_array, [1]), torch.flip(self.t_array, [1])).reshape((-1,))


class GuidedModel:
    """What model_wrapper returns: callable like the reference's `model_fn(x, t_continuous) -> noise`, and — for the fused solver step —
    `raw(x, t)` giving the un-combined network output with its branch count."""

    def __init__(self, model, noise_schedule, model_type, model_kwargs, condition, unconditional_condition, guidance_scale, guided):
        self.model, self.ns, self.model_type, self.model_kwargs = model, noise_schedule, model_type, model_kwargs
        self.condition, self.unconditional_condition, self.guidance_scale, self.guided = condition, unconditional_condition, guidance_scale, guided

    def model_time(self, t_continuous):
        """:246-255: continuous t in [1/N, 1] -> the discrete-time label in [0, 1000 (N-1)/N], as a float."""
        return (t_continuous - 1. / self.ns.total_N) * 1000.

    @staticmethod
    def _cat(u, c):
        if isinstance(c, dict):
            return {k: GuidedModel._cat(u[k], c[k]) for k in c}
        if isinstance(c, (list, tuple)):
            return [GuidedModel._cat(a, b) for a, b in zip(u, c)]
        return torch.cat([u, c])

    def raw(self, x, t_continuous):
        """(network output [branches*B, ...] in batch order [uncond, cond], branches)."""
        t = t_continuous.to(x.device).float().reshape(-1)
        if t.shape[0] == 1:
            t = t.expand(x.shape[0])
        t_in = self.model_time(t)
        if not self.guided:
            return self.model(x, t_in, **self.model_kwargs), 1
        if self.guidance_scale == 1. or self.unconditional_condition is None:
            return self.model(x, t_in, self.condition, **self.model_kwargs), 1
        out = self.model(torch.cat([x] * 2), torch.cat([t_in] * 2), self._cat(self.unconditional_condition, self.condition),
                         **self.model_kwargs)
        return out, 2

    def __call__(self, x, t_continuous):
        out, branches = self.raw(x, t_continuous)
        t = t_continuous.reshape(-1)[:1]
        e, _ = ops.dpm_multistep(x.float().contiguous(), out.float().contiguous(), branches, self.guidance_scale,
                                 float(self.ns.marginal_std(t)), float(self.ns.marginal_alpha(t)), predict_x0=False,
                                 v_param=self.model_type == "v")
        return e


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", condition=None,
                  unconditional_condition=None, guidance_scale=1., classifier_fn=None, classifier_kwargs={}):
    """:161-316."""
    assert model_type in ["noise", "x_start", "v"]
    assert guidance_type in ["uncond", "classifier", "classifier-free"]
    if model_type == "x_start":
        raise NotImplementedError("model_wrapper: x_start-prediction networks are not on the AnyEdit path (sampler.py:4-7)")
    if guidance_type == "classifier":
        raise NotImplementedError("model_wrapper: classifier guidance is not on the AnyEdit path (sampler.py:73)")
    return GuidedModel(model, noise_schedule, model_type, dict(model_kwargs), condition, unconditional_condition, guidance_scale,
                       guided=guidance_type == "classifier-free")


class DPM_Solver:
    """:319-1101, multistep orders 1-2."""

    def __init__(self, model_fn, noise_schedule, predict_x0=False, thresholding=False, max_val=1.):
        if thresholding:
            raise NotImplementedError("DPM_Solver: dynamic thresholding is for pixel-space models; DPMSolverSampler passes False (sampler.py:81)")
        self.model = model_fn
        self.noise_schedule = noise_schedule
        self.predict_x0 = predict_x0
        self.thresholding = thresholding
        self.max_val = max_val

    # ---- host-side scalars
    def get_time_steps(self, skip_type, t_T, t_0, N, device):
        """:376-403 (the times stay on the host)."""
        if skip_type == 'logSNR':
            lambda_T = self.noise_schedule.marginal_lambda(torch.tensor([t_T]))
            lambda_0 = self.noise_schedule.marginal_lambda(torch.tensor([t_0]))
            return self.noise_schedule.inverse_lambda(torch.linspace(lambda_T.item(), lambda_0.item(), N + 1))
        elif skip_type == 'time_uniform':
            return torch.linspace(t_T, t_0, N + 1)
        elif skip_type == 'time_quadratic':
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'".format(skip_type))

    def _update_scalars(self, t_prev_1, t_prev_0, t, order, solver_type):
        """(a, b, c, inv_r0) of x_t = a x - b m0 - c inv_r0 (m0 - m1): :469-513 (order 1) and :723-777 (order 2), as fp32 scalars."""
        ns = self.noise_schedule
        lambda_0, lambda_t = ns.marginal_lambda(t_prev_0), ns.marginal_lambda(t)
        h = lambda_t - lambda_0
        if self.predict_x0:
            a = ns.marginal_std(t) / ns.marginal_std(t_prev_0)
            scale_t = torch.exp(ns.marginal_log_mean_coeff(t))
            em = torch.expm1(-h) if order == 1 else torch.exp(-h) - 1.
        else:
            a = torch.exp(ns.marginal_log_mean_coeff(t) - ns.marginal_log_mean_coeff(t_prev_0))
            scale_t = ns.marginal_std(t)
            em = torch.expm1(h) if order == 1 else torch.exp(h) - 1.
        b = scale_t * em
        if order == 1:
            return float(a), float(b), 0.0, 0.0
        r0 = (lambda_0 - ns.marginal_lambda(t_prev_1)) / h
        if solver_type == 'dpm_solver':
            c = 0.5 * b
        elif self.predict_x0:
            c = -(scale_t * (em / h + 1.))
        else:
            c = scale_t * (em / h - 1.)
        return float(a), float(b), float(c), float(1. / r0)

    # ---- device side: one network evaluation + one fused kernel
    def _evaluate(self, x, t, m_prev=None, update=None, predict_x0=None):
        ns = self.noise_schedule
        px0 = self.predict_x0 if predict_x0 is None else predict_x0
        if isinstance(self.model, GuidedModel):
            out, branches = self.model.raw(x, t)
            scale, v = self.model.guidance_scale, self.model.model_type == "v"
        else:
            out, branches, scale, v = self.model(x, t.to(x.device).expand(x.shape[0])), 1, 1.0, False
        return ops.dpm_multistep(x, out.float().contiguous(), branches, scale, float(ns.marginal_std(t)), float(ns.marginal_alpha(t)),
                                 predict_x0=px0, v_param=v, m_prev=m_prev, update=update)

    def noise_prediction_fn(self, x, t):
        return self._evaluate(x.float().contiguous(), t.reshape(-1)[:1].cpu(), predict_x0=False)[0]

    def data_prediction_fn(self, x, t):
        """:352-365."""
        return self._evaluate(x.float().contiguous(), t.reshape(-1)[:1].cpu(), predict_x0=True)[0]

    def model_fn(self, x, t):
        return self.data_prediction_fn(x, t) if self.predict_x0 else self.noise_prediction_fn(x, t)

    def denoise_to_zero_fn(self, x, s):
        return self.data_prediction_fn(x, s)

    def sample(self, x, steps=20, t_start=None, t_end=None, order=3, skip_type='time_uniform', method='singlestep',
               lower_order_final=True, denoise_to_zero=False, solver_type='dpm_solver', atol=0.0078, rtol=0.05):
        """:939-1101 for method='multistep': S network evaluations for S steps; evaluation i yields the history value m_i and, in the same
        kernel, x at the next step time."""
        if method != 'multistep':
            raise NotImplementedError(f"DPM_Solver.sample: method '{method}' is not on the AnyEdit path (sampler.py:82 uses 'multistep')")
        if order not in (1, 2):
            raise NotImplementedError("DPM_Solver.sample: multistep orders 1 and 2 are implemented (sampler.py:82 uses order 2)")
        if solver_type not in ['dpm_solver', 'taylor']:
            raise ValueError("'solver_type' must be either 'dpm_solver' or 'taylor', got {}".format(solver_type))
        t_0 = 1. / self.noise_schedule.total_N if t_end is None else t_end
        t_T = self.noise_schedule.T if t_start is None else t_start
        assert steps >= order
        ts = self.get_time_steps(skip_type=skip_type, t_T=t_T, t_0=t_0, N=steps, device=x.device)
        assert ts.shape[0] - 1 == steps
        x = x.float().contiguous()
        m_prev = None
        with torch.no_grad():
            for i in range(steps):
                step = i + 1                                            # the reference's loop index of the update to ts[step]
                if i == 0 or order == 1:
                    step_order = 1                                      # :1048-1054 start-up with lower orders
                elif lower_order_final and steps < 15:
                    step_order = min(order, steps + 1 - step)           # :1057-1058
                else:
                    step_order = order
                upd = self._update_scalars(ts[i - 1:i] if i > 0 else None, ts[i:i + 1], ts[step:step + 1], step_order, solver_type)
                m_prev, x = self._evaluate(x, ts[i:i + 1], m_prev=m_prev if step_order == 2 else None, update=upd)
            if denoise_to_zero:
                x = self.denoise_to_zero_fn(x, torch.ones((1,)) * t_0)
        return x
