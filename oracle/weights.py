"""Version-independent synthetic weights / inputs (TEST INFRASTRUCTURE).

There are no checkpoints in the build or GPU containers (SURVEY.md section 7
"Hard parts"), so parity and perf runs use seeded synthetic parameters with the
checkpoint's key names and shapes.  The generator is counter-based integer
arithmetic (splitmix64) followed by exact float64 operations, so the very same
numbers come out on every machine, torch version and device -- unlike
`torch.manual_seed` streams.

Zero-initialised layers of the reference (`zero_module`,
ldm/modules/diffusionmodules/util.py:174, used at openaimodel.py:229,836 and
attention.py:218-223,296) are deliberately filled with non-zero values: with
the reference's own init the UNet output is identically zero and parity would
be vacuous.
"""
import zlib

import numpy as np
import torch

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + _GOLDEN
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def unit_noise(name, shape, seed=0):
    """Deterministic ~N(0,1)-like noise (Irwin-Hall of 4 uniforms, unit variance).

    float64 result; every operation is exact or correctly rounded IEEE, hence
    bit-reproducible.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode("utf-8"))) ^ (np.uint64(seed) << np.uint64(32))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(4) + _splitmix64(np.array([key], dtype=np.uint64))[0]
    acc = np.zeros(n, dtype=np.float64)
    for j in range(4):
        with np.errstate(over="ignore"):
            bits = _splitmix64(ctr + np.uint64(j))
        acc += (bits >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)
    z = (acc - 2.0) * np.sqrt(3.0)
    return z.reshape(shape)


def synth_param(name, shape, seed=0):
    """Synthetic value for a parameter called `name` (float32 numpy array)."""
    shape = tuple(int(s) for s in shape)
    z = unit_noise(name, shape, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = z / np.sqrt(fan_in)
    elif leaf == "weight":            # GroupNorm / LayerNorm scale
        v = 1.0 + 0.1 * z
    elif leaf == "bias":
        v = 0.05 * z
    else:                              # learnable_vector, bbox_uncond_vector, ...
        v = z
    return v.astype(np.float32)


def synth_state_dict(shapes, seed=0):
    """`shapes`: {key: shape}.  Returns {key: float32 torch tensor}."""
    return {k: torch.from_numpy(synth_param(k, s, seed)) for k, s in shapes.items()}


def fill_module_(module, seed=0, prefix=""):
    """Overwrite every floating parameter of an nn.Module in place (used on the
    reference modules by tests/golden/make_golden.py and on the product modules)."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(torch.from_numpy(synth_param(prefix + k, p.shape, seed)).to(p.dtype))
    return module


def synth_input(name, shape, seed=0, kind="normal"):
    z = unit_noise("input:" + name, tuple(shape), seed)
    if kind == "uniform":             # ~U(-1,1)-ish, clipped
        z = np.clip(z / 1.7320508075688772, -1.0, 1.0)
    return torch.from_numpy(z.astype(np.float32))
