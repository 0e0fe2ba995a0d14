"""Schedules (host, float64 / int64) and parameter holders for the engine.

Mirrors the names of the reference's ldm/modules/diffusionmodules/util.py:
make_beta_schedule :21, make_ddim_timesteps :46, make_ddim_sampling_parameters :63,
extract_into_tensor :96, timestep_embedding :151, normalization :199, conv_nd :218,
linear :231, zero_module :174.

The holders (`Conv2d`, `Linear`, `GroupNorm32`, `LayerNorm`) keep fp32 master
parameters under the reference's `state_dict` keys and hand the engine
pre-packed device copies; they have no `forward` of their own -- the arithmetic
lives in the HIP kernels the owning block calls.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .... import engine_dtype, ops


# --------------------------------------------------------------------------------------
# schedules
# --------------------------------------------------------------------------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise ValueError(f"schedule '{schedule}' is not used by MObI's configs")
    # torch.linspace (FMA-based on CPU) -- bit-compatible with the reference's table
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method != "uniform":
        raise NotImplementedError(f'ddim discretization "{ddim_discr_method}" is not used by MObI')
    c = num_ddpm_timesteps // num_ddim_timesteps
    steps_out = np.arange(0, num_ddpm_timesteps, c, dtype=np.int64) + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """Returns (sigmas f64, alphas f32, alphas_prev f64) with the reference's mixed
    float32 / float64 rounding (see oracle/schedule.py for the derivation; pinned by
    tests/golden/schedule_tables.npz)."""
    ac = np.asarray(alphacums.detach().cpu().numpy() if isinstance(alphacums, torch.Tensor) else alphacums,
                    dtype=np.float32)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist())
    recip = (np.float32(1) / (np.float32(1) - alphas)).astype(np.float64)
    sigmas = eta * np.sqrt(recip * (1 - alphas_prev) * (1 - alphas.astype(np.float64) / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the sigma_t schedule {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def timestep_freqs(dim, max_period=10000):
    """fp32 frequency table, computed exactly as the reference does on the host."""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


_FREQ_CACHE = {}


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    if repeat_only or dim % 2:
        raise NotImplementedError("only the even-dim sinusoidal form is used by MObI")
    key = (dim, max_period, timesteps.device)
    if key not in _FREQ_CACHE:
        _FREQ_CACHE[key] = timestep_freqs(dim, max_period).to(timesteps.device)
    return ops.timestep_embedding(timesteps.to(torch.int64).contiguous(), _FREQ_CACHE[key])


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


# --------------------------------------------------------------------------------------
# parameter holders
# --------------------------------------------------------------------------------------
import os as _os

# measured on MI355X (tools/kbench.py): the chunk-major k order is 10-15 % SLOWER than tap-major for the UNet's 3x3
# convolutions (per-tile re-derivation of the tap window outweighs the L2 reuse) -> off by default, kept for study
_CHUNK_MAJOR = _os.environ.get("MOBI_CHUNK_MAJOR", "0") == "1"


# bumped whenever parameters are replaced wholesale (load_state_dict, .to() / .cuda() / .float()): captured step
# graphs (mobi_amd/graph.py) hold the addresses of the packed copies and are dropped when this moves
WEIGHTS_EPOCH = [0]


class _Holder(nn.Module):
    """Caches device-side packed copies keyed on (dtype, device, parameter versions)."""

    def _apply(self, fn, *args, **kwargs):
        WEIGHTS_EPOCH[0] += 1
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        WEIGHTS_EPOCH[0] += 1
        return super()._load_from_state_dict(*args, **kwargs)

    def _cached(self, tag, build):
        ps = list(self.parameters(recurse=False))
        key = (tag, engine_dtype(), ps[0].device, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))
        cache = self.__dict__.setdefault("_pack_cache", {})
        hit = cache.get(tag)
        if hit is None or hit[0] != key:
            hit = (key, build())
            cache[tag] = hit
        return hit[1]

    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; its arithmetic runs inside the "
                           "HIP kernels of the owning block")


class Conv2d(_Holder):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = \
            in_channels, out_channels, ks, stride, pd
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *ks))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * ks[0] * ks[1])
        nn.init.uniform_(self.bias, -bound, bound)

    def packed(self):
        """T [O][kh*kw*I] for the matrix-core / small-cout kernels."""
        return self._cached("mfma", lambda: ops.pack_conv(self.weight, self.bias, engine_dtype(), self.weight.device,
                                                          chunk_major=_CHUNK_MAJOR))

    def packed_tap_major(self):
        """k = tap*C + c order (the small-cout direct kernel)."""
        return self._cached("tapmajor", lambda: ops.pack_conv(self.weight, self.bias, engine_dtype(),
                                                              self.weight.device))

    def packed_dup(self):
        """The weights duplicated along the input channels ([W ; W]): the convolution of a `hi | lo` pair of a tensor
        (ops.groupnorm(..., out_mode=GN_OUT_SPLIT)) is then W hi + W lo = W y with y to ~22 bits in one launch."""
        return self._cached("dup", lambda: ops.pack_conv(torch.cat([self.weight, self.weight], dim=1), self.bias, engine_dtype(),
                                                         self.weight.device))

    def packed_dup3(self):
        """[W ; W ; W - round(W)]: against a `hi | lo | hi` operand the launch computes round(W) hi + round(W) lo + (W - round(W)) hi --
        activations AND weights to ~22 bits (the hi x lo-of-W term; lo x lo-of-W is below fp32's own rounding)."""
        def build():
            w = self.weight.detach().float()
            wlo = w - w.to(engine_dtype()).float()
            return ops.pack_conv(torch.cat([w, w, wlo], dim=1), self.bias, engine_dtype(), self.weight.device)
        return self._cached("dup3", build)

    def packed_thin(self):
        """as packed(), with the (< 32) input channels zero-padded to 32 for ops.pack_sources inputs."""
        return self._cached("thin", lambda: ops.pack_conv_padded_cin(self.weight, self.bias, engine_dtype(),
                                                                     self.weight.device))

    def packed_f32(self):
        """fp32 [O][I*kh*kw] (OIHW flattened) for the small-cin direct kernel."""
        return self._cached("f32", lambda: (self.weight.detach().float().reshape(self.out_channels, -1).contiguous(),
                                            self.bias.detach().float().contiguous()))


class Linear(_Holder):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
            bound = 1 / math.sqrt(in_features)
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)

    def packed(self):
        return self._cached("mfma", lambda: ops.pack_linear(self.weight, self.bias, engine_dtype(), self.weight.device))

    def packed_geglu(self):
        return self._cached("geglu", lambda: ops.pack_geglu(self.weight, self.bias, engine_dtype(), self.weight.device))

    def skinny(self):
        """(T weight [n][k], fp32 bias) for mobi_skinny_linear."""
        return self._cached("skinny", lambda: (
            self.weight.detach().to(engine_dtype()).contiguous(),
            None if self.bias is None else self.bias.detach().float().contiguous()))


class _Norm(_Holder):
    def affine(self):
        return self._cached("affine", lambda: (self.weight.detach().float().contiguous(),
                                               self.bias.detach().float().contiguous()))


class GroupNorm32(_Norm):
    """32 groups, fp32 statistics (reference: GroupNorm32, util.py:214-216)."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        assert num_groups == 32
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))


class LayerNorm(_Norm):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.normalized_shape, self.eps = (dim,), eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class Marker(nn.Module):
    """Parameter-free placeholder that keeps the reference's Sequential indices
    (nn.SiLU / nn.Dropout / nn.Identity positions) so `state_dict` keys line up."""

    def forward(self, x):
        return x


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims}")
    return Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


# --------------------------------------------------------------------------------------
# layout helpers at the operator boundary
# --------------------------------------------------------------------------------------
def enter(x):
    """fp32 NCHW (reference layout) -> engine layout; engine tensors pass through.
    Returns (tensor, was_external)."""
    if x.dtype == torch.float32:
        return ops.to_nhwc(x.contiguous(), engine_dtype()), True
    return x, False


def leave(x, external):
    return ops.to_nchw_f32(x.contiguous()) if external else x
