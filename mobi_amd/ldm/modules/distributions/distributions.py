"""DiagonalGaussianDistribution on the engine (reference:
ldm/modules/distributions/distributions.py:24-62).  Holds the fp32 NCHW moments; `sample`
draws its noise from torch's CPU generator and moves it to the device exactly as the
reference does (:36), or takes it explicitly for parity runs."""
import torch

from .... import ops


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic

    @property
    def mean(self):
        return torch.chunk(self.parameters, 2, dim=1)[0]

    def sample(self, noise=None, scale=1.0, out=None, c_off=0):
        """z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise); optionally written into
        channels [c_off, c_off + c) of `out` (the 9-channel UNet input, ddpm.py:1021)."""
        b, c2, h, w = self.parameters.shape
        c = c2 // 2
        if noise is None:
            noise = torch.randn((b, c, h, w)).to(device=self.parameters.device)
        if self.deterministic:
            noise = torch.zeros_like(noise)
        if out is None:
            out = torch.empty((b, c, h, w), device=self.parameters.device, dtype=torch.float32)
        return ops.posterior_sample(self.parameters.contiguous(), noise.float().contiguous(), out, c_off, scale)

    def mode(self):
        return self.mean
