"""mobi_amd -- MI355X (gfx950) native denoising engine behind MObI's `ldm` operator API.

`mobi_amd.ldm.*` mirrors the module paths, class names, constructor kwargs and
`state_dict` keys of the reference's `ldm.*` for the sampling path
(LatentDiffusion, DDIMSampler / PLMSSampler, UNetModel, AutoencoderKL); all the
arithmetic runs in hand-written HIP kernels reached through the C ABI of
`include/mobi_engine.h` (`libmobi_hip.so`).  There is no CPU or PyTorch-op
fallback: without the library, or with tensors off the GPU, calls raise.
"""
import torch

_ENGINE_DTYPE = torch.bfloat16


def set_engine_dtype(dtype):
    """Storage type of activations and matrix-core operands (float16 | bfloat16);
    accumulation and normalisation statistics are always fp32."""
    global _ENGINE_DTYPE
    if dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("engine dtype must be torch.float16 or torch.bfloat16")
    _ENGINE_DTYPE = dtype


def engine_dtype():
    return _ENGINE_DTYPE
