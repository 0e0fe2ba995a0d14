"""DDIM / PLMS sampling loops on CPU (TEST INFRASTRUCTURE).

Reference: ldm/models/diffusion/ddim.py (make_schedule :25-54, ddim_sampling
:115-163, p_sample_ddim :166-213), ldm/models/diffusion/plms.py
(plms_sampling :116-170, p_sample_plms :174-239), DDPM.q_sample
(ldm/models/diffusion/ddpm.py:284-287).

`eps_fn(x, t, cond)` stands for `LatentDiffusion.apply_model`
(ddpm.py:1060-1157 -> DiffusionWrapper 'crossattn' :1709-1711).
All noise is passed in explicitly (`x_T`, `step_noise`): the reference draws it
from device RNG streams that cannot be reproduced.
"""
import numpy as np
import torch

from . import schedule


class Schedule:
    """Tables a sampler needs: the model's float32 DDPM buffers
    (ddpm.py:143-153) and the DDIM subset tables (ddim.py:25-50)."""

    def __init__(self, S, eta=0.0, n_timestep=1000, linear_start=0.00085, linear_end=0.012):
        self.buffers = schedule.ddpm_buffers(n_timestep, linear_start, linear_end)
        self.timesteps = schedule.ddim_timesteps(S, n_timestep)
        p = schedule.ddim_parameters(self.buffers["alphas_cumprod"], self.timesteps, eta)
        self.alphas, self.alphas_prev = p["alphas"], p["alphas_prev"]
        self.sigmas, self.sqrt_one_minus_alphas = p["sigmas"], p["sqrt_one_minus_alphas"]


def _coef(v, b):
    # `torch.full((b,1,1,1), table[index])` -> float32 (ddim.py:195-198)
    return torch.full((b, 1, 1, 1), float(v), dtype=torch.float32)


def q_sample(buffers, x0, t, noise):
    """ddpm.py:284-287 with util.extract_into_tensor (util.py:96-99)."""
    a = torch.from_numpy(buffers["sqrt_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    s = torch.from_numpy(buffers["sqrt_one_minus_alphas_cumprod"])[t].reshape(-1, 1, 1, 1)
    return a * x0 + s * noise


def _model_eps(eps_fn, img, rest, t, cond, scale, uncond):
    """ddim.py:168-184: channel concat, then plain or classifier-free call."""
    x = torch.cat([img, rest], dim=1)
    if uncond is None or scale == 1.0:
        return eps_fn(x, t, cond)
    e_u, e_c = eps_fn(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uncond, cond])).chunk(2)
    return e_u + scale * (e_c - e_u)


def _x_prev(sch, index, x, e_t, noise, temperature=1.0):
    """ddim.py:195-213 / plms.py:199-214."""
    b = x.shape[0]
    a_t, a_prev = _coef(sch.alphas[index], b), _coef(sch.alphas_prev[index], b)
    sigma_t, s1m = _coef(sch.sigmas[index], b), _coef(sch.sqrt_one_minus_alphas[index], b)
    pred_x0 = (x - s1m * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + sigma_t * noise * temperature
    return x_prev, pred_x0


def ddim_sample(eps_fn, sch: Schedule, cond, x_T, rest, scale=1.0, uncond=None,
                mask=None, x0=None, mask_noise=None, step_noise=None, log_every_t=100):
    """DDIMSampler.ddim_sampling.  `rest` = cat[inpaint_image, inpaint_mask]
    (the `test_model_kwargs` of ddim.py:168-170).  Returns (samples, intermediates)."""
    img = x_T
    b = img.shape[0]
    total = sch.timesteps.shape[0]
    inter = {"x_inter": [img], "pred_x0": [img], "ts": []}
    for i, step in enumerate(np.flip(sch.timesteps)):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        inter["ts"].append(int(step))
        if mask is not None:                                           # ddim.py:145-148
            img_orig = q_sample(sch.buffers, x0, ts, mask_noise[i])
            img = img_orig * mask + (1.0 - mask) * img
        e_t = _model_eps(eps_fn, img, rest, ts, cond, scale, uncond)
        noise = step_noise[i] if step_noise is not None else torch.zeros_like(img)
        img, pred_x0 = _x_prev(sch, index, img, e_t, noise)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter


def plms_sample(eps_fn, sch: Schedule, cond, x_T, rest, scale=1.0, uncond=None, log_every_t=100):
    """PLMSSampler.plms_sampling + p_sample_plms (eta must be 0, plms.py:25-26)."""
    img = x_T
    b = img.shape[0]
    time_range = np.flip(sch.timesteps)
    total = sch.timesteps.shape[0]
    inter = {"x_inter": [img], "pred_x0": [img]}
    old_eps = []
    zero = torch.zeros_like(img)
    for i, step in enumerate(time_range):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = _model_eps(eps_fn, img, rest, ts, cond, scale, uncond)
        if len(old_eps) == 0:                                          # pseudo improved Euler, plms.py:219-224
            x_prev, _ = _x_prev(sch, index, img, e_t, zero)
            e_next = _model_eps(eps_fn, x_prev, rest, ts_next, cond, scale, uncond)
            e_prime = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = _x_prev(sch, index, img, e_prime, zero)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
    return img, inter
