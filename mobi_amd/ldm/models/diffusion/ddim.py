"""DDIM sampler on the engine (reference: ldm/models/diffusion/ddim.py -- make_schedule :25-54,
sample :57-112, ddim_sampling :115-163, p_sample_ddim :166-213).  Same call signature; the
latent state and every coefficient stay fp32; the per-step arithmetic is one HIP kernel
(mobi_ddim_step) and the channel concat feeding the UNet is never materialised."""
import numpy as np
import torch

from .... import graph, ops
from ...modules.diffusionmodules.util import (make_ddim_sampling_parameters, make_ddim_timesteps, noise_like)


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = bool(kwargs.get("graph", True))      # one HIP graph launch per step (mobi_amd/graph.py)

    def register_buffer(self, name, attr):
        # the reference pins buffers to "cuda" (ddim.py:19-23); follow the model's device instead
        if isinstance(attr, torch.Tensor):
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize,
                                                  num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        to_torch = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        ac = alphas_cumprod.detach().cpu()
        self.register_buffer("betas", to_torch(self.model.betas))
        self.register_buffer("alphas_cumprod", to_torch(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", to_torch(self.model.alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", to_torch(torch.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", to_torch(torch.sqrt(1. - ac)))
        self.register_buffer("log_one_minus_alphas_cumprod", to_torch(torch.log(1. - ac)))
        self.register_buffer("sqrt_recip_alphas_cumprod", to_torch(torch.sqrt(1. / ac)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", to_torch(torch.sqrt(1. / ac - 1)))
        sig, a, a_prev = make_ddim_sampling_parameters(alphacums=ac, ddim_timesteps=self.ddim_timesteps,
                                                       eta=ddim_eta, verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(np.float32(1) - a)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        if quantize_x0 or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("quantize_x0 / score_corrector / noise_dropout are not on MObI's path")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.ddim_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback,
                                  mask=mask, x0=x0, temperature=temperature, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, timesteps=None, mask=None, x0=None,
                      img_callback=None, log_every_t=100, temperature=1., unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, mask_noise=None, step_noise=None, **kwargs):
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        img = img.contiguous()
        if timesteps is not None:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            steps = self.ddim_timesteps[:subset_end]
        else:
            steps = self.ddim_timesteps
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(steps)
        total_steps = steps.shape[0]
        self._weights_fp = graph.weights_fingerprint(self.model)
        graphed = self.use_graph and graph.usable(img) and isinstance(cond, torch.Tensor)
        keep = (lambda t_: t_.clone()) if graphed else (lambda t_: t_)     # graph outputs are overwritten next step
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                nz = mask_noise[i] if mask_noise is not None else torch.randn_like(x0)
                img = ops.mask_blend_(img.clone(), x0.float().contiguous(), nz.float().contiguous(),
                                      mask.float().contiguous(),
                                      float(self.sqrt_alphas_cumprod[int(step)]),
                                      float(self.sqrt_one_minus_alphas_cumprod[int(step)]))
            nz = None
            if float(self.ddim_sigmas[index]) != 0.0:
                nz = step_noise[i] if step_noise is not None else noise_like(img.shape, device)
            img, pred_x0 = self._step(img, cond, ts, index, temperature, unconditional_guidance_scale,
                                      unconditional_conditioning, nz, kwargs, step_value=int(step))
            if callback:
                callback(i)
            if img_callback:
                img_callback(keep(pred_x0), i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(keep(img))
                intermediates["pred_x0"].append(keep(pred_x0))
        return keep(img), intermediates

    def _eps(self, x, c, t, unconditional_guidance_scale, unconditional_conditioning, kwargs, cfg_ctx=None):
        """Returns (e_cond, e_uncond|None): the classifier-free mix happens in mobi_ddim_step.
        cfg_ctx: the caller's own [uncond ; cond] token tensor (a captured step keeps one static buffer)."""
        if "test_model_kwargs" in kwargs:
            kw = kwargs["test_model_kwargs"]
            parts = [x, kw["inpaint_image"], kw["inpaint_mask"]]
        elif "rest" in kwargs:
            parts = [x, kwargs["rest"]]
        else:
            raise Exception("kwargs must contain either 'test_model_kwargs' or 'rest' key")
        parts = [p.float().contiguous() for p in parts]
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            return self.model.apply_model(parts, t, c), None
        parts2 = [torch.cat([p] * 2) for p in parts]
        # [uncond ; cond] tokens are the same tensor at every step of a run: build it once so that the UNet's
        # per-context cache (attn2 vector, adapter k/v) hits
        if cfg_ctx is None:
            ck = (id(unconditional_conditioning), id(c), unconditional_conditioning._version, c._version)
            if getattr(self, "_cfg_ctx_key", None) != ck:
                self._cfg_ctx_key, self._cfg_ctx_refs = ck, (unconditional_conditioning, c)
                self._cfg_ctx = torch.cat([unconditional_conditioning, c])
            cfg_ctx = self._cfg_ctx
        out = self.model.apply_model(parts2, torch.cat([t] * 2), cfg_ctx)
        e_uncond, e_cond = out.chunk(2)
        return e_cond.contiguous(), e_uncond.contiguous()

    def _coef_table(self):
        """fp32 device table [S, 4] = {a_t, a_prev, sigma_t, sqrt(1 - a_t)} of the current schedule: exactly the
        values `float(...)` of the numpy tables become when passed by value (ctypes c_float rounds the same way)."""
        key = (id(self.ddim_alphas), id(self.ddim_sigmas), self.model.betas.device)
        if getattr(self, "_coef_key", None) != key:
            tab = np.stack([np.asarray(self.ddim_alphas, dtype=np.float64), np.asarray(self.ddim_alphas_prev, np.float64),
                            np.asarray(self.ddim_sigmas, np.float64),
                            np.asarray(self.ddim_sqrt_one_minus_alphas, np.float64)], axis=1).astype(np.float32)
            self._coef_key, self._coef_dev = key, torch.from_numpy(tab).to(self.model.betas.device)
        return self._coef_dev

    def refresh_weights_fingerprint(self):
        from ...modules.diffusionmodules.util import WEIGHTS_EPOCH
        self._weights_fp, self._weights_fp_epoch = graph.weights_fingerprint(self.model), WEIGHTS_EPOCH[0]

    def _step(self, x, c, t, index, temperature, scale, uncond, noise, kwargs, step_value=None):
        """One denoising step -> (x_prev, pred_x0).  On the GPU the step is ONE graph launch (mobi_amd/graph.py):
        the returned tensors are then the graph's static outputs, valid until the next step."""
        sigma = float(self.ddim_sigmas[index])
        if noise is None and sigma != 0.0:       # the reference draws at every step (ddim.py:209); it matters here only
            noise = noise_like(x.shape, x.device)
        use_graph = self.use_graph and graph.usable(x) and isinstance(c, torch.Tensor)
        if use_graph and step_value is None:
            # called from outside the sampling loops (ddim_sampling / plms_sampling pass the value they already have and
            # refresh the weights fingerprint once per run): one read-back of `t`; the captured step takes ONE timestep for
            # the whole batch, so a ragged `t` stays on the eager path, which honours it; in-place parameter edits since
            # the last call move the fingerprint and re-capture
            tv = t.tolist()
            if len(set(tv)) == 1:
                step_value = int(tv[0])
                # the fingerprint is a Python walk over every parameter: once per weights epoch here (load_state_dict /
                # .to() bump it), not per step -- an external loop that edits parameters in place between two steps calls
                # `refresh_weights_fingerprint()` (the sampling loops of this class refresh it once per run)
                from ...modules.diffusionmodules.util import WEIGHTS_EPOCH
                if getattr(self, "_weights_fp_epoch", None) != WEIGHTS_EPOCH[0] or getattr(self, "_weights_fp", None) is None:
                    self.refresh_weights_fingerprint()
            else:
                use_graph = False
        if use_graph:
            g = graph.get(self, "ddim", x, c, uncond, scale, kwargs, temperature=temperature, has_noise=sigma != 0.0)
            self._last_step_was_graph = True
            return g.run(x, step_value, self._coef_table()[index], noise)
        self._last_step_was_graph = False
        e_cond, e_uncond = self._eps(x, c, t, scale, uncond, kwargs)
        x_prev, pred_x0, _ = ops.ddim_step(
            x, e_cond, e_uncond=e_uncond, noise=noise, cfg_scale=float(scale),
            a_t=float(self.ddim_alphas[index]), a_prev=float(self.ddim_alphas_prev[index]), sigma_t=sigma,
            sqrt_one_minus_at=float(self.ddim_sqrt_one_minus_alphas[index]), temperature=float(temperature))
        return x_prev, pred_x0

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, temperature=1., unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, noise=None, step_value=None, **kwargs):
        """ddim.py:166-213 of the reference.  `step_value`: the (batch-uniform) timestep as a Python int when the
        caller has it -- saves reading it back from `t`."""
        x = x.float().contiguous()
        x_prev, pred_x0 = self._step(x, c, t, index, temperature, unconditional_guidance_scale,
                                     unconditional_conditioning, noise, kwargs, step_value)
        if getattr(self, "_last_step_was_graph", False):
            return x_prev.clone(), pred_x0.clone()       # static graph outputs: hand out copies
        return x_prev, pred_x0
