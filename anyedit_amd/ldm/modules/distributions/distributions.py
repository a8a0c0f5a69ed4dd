"""ldm/modules/distributions/distributions.py:24-62 — DiagonalGaussianDistribution on the HIP path (ae_gaussian_moments_f32)."""
import numpy as np
import torch

from anyedit_amd import ops


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic
        _, self.mean, self.logvar, self.std = ops.gaussian_moments(parameters, None, want_stats=True)  # logvar clamped to [-30, 20]
        self.var = self.std * self.std
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)
        self.randn = torch.randn  # tests may replace this to replay a CPU noise stream

    def sample(self):
        """distributions.py:35-37: mean + std * N(0, I)."""
        if self.deterministic:
            return self.mean
        noise = self.randn(self.mean.shape, device=self.parameters.device)
        return ops.gaussian_moments(self.parameters, noise)

    def mode(self):
        return self.mean

    def kl(self, other=None):
        """distributions.py:39-51 (training-time regulariser of the first stage: host-side reduction of small tensors)."""
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar,
                               dim=[1, 2, 3])

    def nll(self, sample, dims=[1, 2, 3]):
        if self.deterministic:
            return torch.Tensor([0.])
        logtwopi = np.log(2.0 * np.pi)
        return 0.5 * torch.sum(logtwopi + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=dims)
