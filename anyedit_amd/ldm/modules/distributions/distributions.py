"""DiagonalGaussianDistribution (ldm/modules/distributions/distributions.py:24-62) over the HIP moments kernel: one launch of
`ae_gaussian_moments_f32` splits the encoder's `[B, 2z, h, w]` moments into mean / clamped log-variance / std (and, for `sample`, adds
the noise) — the reference's chunk + clamp + two exps + multiply-add as one pass."""
import math

import torch

from anyedit_amd import ops

_SPATIAL = (1, 2, 3)


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic
        _, self.mean, self.logvar, self.std = ops.gaussian_moments(parameters, None, want_stats=True)  # logvar clamped to [-30, 20]
        if deterministic:
            self.std = torch.zeros_like(self.mean)
            self.var = self.std
        else:
            self.var = self.std * self.std
        self.randn = torch.randn  # tests may replace this to replay a CPU noise stream

    def mode(self):
        return self.mean

    def sample(self):
        """mean + std * N(0, I) (distributions.py:35-37); the mean itself when deterministic."""
        if self.deterministic:
            return self.mean
        return ops.gaussian_moments(self.parameters, self.randn(self.mean.shape, device=self.parameters.device))

    # The two scalars below belong to the first stage's own training (distributions.py:39-62), not to the edit path: small host-side torch
    # reductions over tensors the kernel already produced.
    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:                                          # against N(0, I)
            terms = torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar
        else:
            terms = torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar
        return 0.5 * torch.sum(terms, dim=_SPATIAL)

    def nll(self, sample, dims=_SPATIAL):
        if self.deterministic:
            return torch.Tensor([0.])
        return 0.5 * torch.sum(math.log(2.0 * math.pi) + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=list(dims))
