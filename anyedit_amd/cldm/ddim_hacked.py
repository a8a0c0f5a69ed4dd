"""Mirror of AnyEdit_Collection/other_modules/cldm/ddim_hacked.py — the DDIM sampler of the ControlLDM / AnyDoor path
(visual_reference_tool.py builds `DDIMSampler(model)` from this module).  It differs from ldm's sampler in two places:
  * guidance takes TWO network calls, conditional then unconditional (:189-193), because the unconditional conditioning of a
    ControlLDM may drop the control branch (`c_concat: None`), which a single concatenated batch cannot express;
  * the inversion loop queries the network at `ddim_timesteps[i]` (:237-254), not at the loop index.
Everything else — schedule, integer bookkeeping, fused update kernels — is inherited from the ldm mirror.
"""
import numpy as np
import torch

from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler as _LdmDDIMSampler


class DDIMSampler(_LdmDDIMSampler):
    def _model_eps(self, x, c, t, unconditional_guidance_scale, unconditional_conditioning):
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(x, t, c), 1
        model_t = self.model.apply_model(x, t, c)
        model_uncond = self.model.apply_model(x, t, unconditional_conditioning)
        return torch.cat([model_uncond, model_t]), 2          # batch order [uncond, cond] of the fused guidance + update kernel

    def _encode_timestep(self, i, use_original_steps):
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        return timesteps[i]
