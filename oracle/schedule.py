"""Noise schedules and DDIM tables (TEST INFRASTRUCTURE; numpy int64 / float64).

The integer timestep tables are the bit-exact part of the path (SURVEY.md 8(a)
rows A1-A2).
"""
import numpy as np
import torch


def linear_betas(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """`make_beta_schedule("linear")`, ldm/modules/diffusionmodules/util.py:21-26:
    `torch.linspace` over sqrt(beta) in float64, squared.  torch's CPU linspace
    is FMA-based and differs from `np.linspace` in the last bit for ~15 % of the
    entries, so the restatement calls the same torch routine."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def ddpm_buffers(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """`DDPM.register_schedule`, ldm/models/diffusion/ddpm.py:127-158: float64
    tables, every registered buffer cast to float32."""
    betas = linear_betas(n_timestep, linear_start, linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
    }


def ddim_timesteps(num_ddim, num_ddpm=1000):
    """`make_ddim_timesteps("uniform")`, util.py:46-60.  c = T // S, then
    arange(0, T, c) + 1.  When T % S != 0 the table is LONGER than S (the
    reference's assert is commented out, util.py:55) -- kept."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c)), dtype=np.int64) + 1


def ddim_parameters(alphas_cumprod_f32, timesteps, eta):
    """`make_ddim_sampling_parameters`, util.py:63-74, as called from
    `DDIMSampler.make_schedule`, ldm/models/diffusion/ddim.py:43-50.

    The reference mixes a float32 torch tensor (`alphas`) with a float64 numpy
    array (`alphas_prev`, built through `.tolist()`); the resulting dtypes and
    rounding are reproduced here in plain numpy and pinned bit-for-bit by
    tests/golden/schedule_tables.npz:
      * alphas                 float32
      * alphas_prev            float64 (float32 values widened)
      * sqrt_one_minus_alphas  float32, sqrt(1 - a) evaluated in float32 (ddim.py:50)
      * sigmas                 float64; `(1 - a_prev) / (1 - a)` goes through
        torch's reflected division = float32 reciprocal of (1 - a) times the
        float64 numerator; `a / a_prev` is a float64 division.
    """
    ac = np.asarray(alphas_cumprod_f32, dtype=np.float32)
    a = ac[timesteps]
    a_prev = np.asarray([ac[0]] + ac[timesteps[:-1]].tolist())          # float64
    recip = (np.float32(1) / (np.float32(1) - a)).astype(np.float64)
    sig = eta * np.sqrt(recip * (1 - a_prev) * (1 - a.astype(np.float64) / a_prev))
    return {"sigmas": sig, "alphas": a, "alphas_prev": a_prev,
            "sqrt_one_minus_alphas": np.sqrt(np.float32(1) - a)}
