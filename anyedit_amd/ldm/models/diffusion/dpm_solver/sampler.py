"""DPMSolverSampler on the HIP path — mirror of ldm/models/diffusion/dpm_solver/sampler.py: DPM-Solver++(2M) over
`model.apply_model` with classifier-free guidance; typically 15-25 network evaluations per image instead of DDIM's 50."""
import torch

from anyedit_amd.ldm.util import warn_conditioning_batch
from .dpm_solver import NoiseScheduleVP, model_wrapper, DPM_Solver

MODEL_TYPES = {"eps": "noise", "v": "v"}


class DPMSolverSampler(object):
    def __init__(self, model, **kwargs):
        super().__init__()
        self.model = model
        self.register_buffer('alphas_cumprod', model.alphas_cumprod.clone().detach().to(torch.float32))

    def register_buffer(self, name, attr):
        # sampler.py:20-24 moves every buffer to "cuda"; the schedule is only read on the host here, so it stays where it is
        setattr(self, name, attr)

    def randn(self, shape, device=None):
        return torch.randn(shape, device=device)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        """sampler.py:26-86: returns (samples, None).  Like the reference, mask / x0 / callbacks are accepted and ignored."""
        warn_conditioning_batch(conditioning, batch_size)
        C, H, W = shape
        size = (batch_size, C, H, W)
        device = self.model.betas.device
        img = self.randn(size, device=device) if x_T is None else x_T
        ns = NoiseScheduleVP('discrete', alphas_cumprod=self.alphas_cumprod)
        model_fn = model_wrapper(lambda x, t, c: self.model.apply_model(x, t, c), ns,
                                 model_type=MODEL_TYPES[getattr(self.model, "parameterization", "eps")],
                                 guidance_type="classifier-free", condition=conditioning,
                                 unconditional_condition=unconditional_conditioning, guidance_scale=unconditional_guidance_scale)
        dpm_solver = DPM_Solver(model_fn, ns, predict_x0=True, thresholding=False)
        x = dpm_solver.sample(img, steps=S, skip_type="time_uniform", method="multistep", order=2, lower_order_final=True)
        return x.to(device), None
