"""PLMS sampler on the engine (reference: ldm/models/diffusion/plms.py -- plms_sampling :116-170,
p_sample_plms :174-239): Adams-Bashforth mixes of the last <= 3 eps (mobi_lincomb4) around the
same fp32 update kernel as DDIM; the first step does two UNet evaluations."""
import numpy as np
import torch

from .... import graph, ops
from .ddim import DDIMSampler


class PLMSSampler(DDIMSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        super().make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, eta=0., x_T=None, log_every_t=100, verbose=True,
               unconditional_guidance_scale=1., unconditional_conditioning=None, callback=None, img_callback=None,
               temperature=1., **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.plms_sampling(conditioning, (batch_size, C, H, W), x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, callback=callback,
                                  img_callback=img_callback, temperature=temperature, **kwargs)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, callback=None, img_callback=None, temperature=1.,
                      inpaint_image=None, inpaint_mask=None, **ignored):
        device = self.model.betas.device
        b = shape[0]
        img = (torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32))
        img = img.contiguous()
        kw = {"test_model_kwargs": {"inpaint_image": inpaint_image, "inpaint_mask": inpaint_mask}}
        time_range = np.flip(self.ddim_timesteps)
        total_steps = self.ddim_timesteps.shape[0]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        old_eps = []

        self._weights_fp = graph.weights_fingerprint(self.model)
        graphed = self.use_graph and graph.usable(img) and isinstance(cond, torch.Tensor)

        def model_eps(x, t, step_value):
            """(e_cond, e_uncond | None); on the GPU one graph launch per UNet evaluation -- its outputs are static
            buffers, consumed by `update` before the next evaluation overwrites them."""
            if graphed:
                g = graph.get(self, "eps", x, cond, unconditional_conditioning, unconditional_guidance_scale, kw)
                return g.run(x, step_value)
            return self._eps(x, cond, t, unconditional_guidance_scale, unconditional_conditioning, kw)

        def update(x, e, index, e_uncond=None, want_e=False):
            return ops.ddim_step(x, e, e_uncond=e_uncond, cfg_scale=float(unconditional_guidance_scale),
                                 a_t=float(self.ddim_alphas[index]), a_prev=float(self.ddim_alphas_prev[index]),
                                 sigma_t=0.0, sqrt_one_minus_at=float(self.ddim_sqrt_one_minus_alphas[index]),
                                 temperature=float(temperature), want_e=want_e)

        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device,
                                 dtype=torch.long)
            step_next = int(time_range[min(i + 1, len(time_range) - 1)])
            e_c, e_u = model_eps(img, ts, int(step))
            # one pass of the update kernel also yields e_t (after the CFG mix)
            x_euler, _, e_t = update(img, e_c, index, e_uncond=e_u, want_e=True)
            if len(old_eps) == 0:
                e_c2, e_u2 = model_eps(x_euler, ts_next, step_next)
                _, _, e_next = update(x_euler, e_c2, index, e_uncond=e_u2, want_e=True)
                e_prime = ops.lincomb4([e_t, e_next], [0.5, 0.5])
            elif len(old_eps) == 1:
                e_prime = ops.lincomb4([e_t, old_eps[-1]], [3 / 2, -1 / 2])
            elif len(old_eps) == 2:
                e_prime = ops.lincomb4([e_t, old_eps[-1], old_eps[-2]], [23 / 12, -16 / 12, 5 / 12])
            else:
                e_prime = ops.lincomb4([e_t, old_eps[-1], old_eps[-2], old_eps[-3]],
                                       [55 / 24, -59 / 24, 37 / 24, -9 / 24])
            img, pred_x0, _ = update(img, e_prime, index)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates
