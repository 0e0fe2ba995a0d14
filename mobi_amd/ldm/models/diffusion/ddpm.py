"""LatentDiffusion (inference subset) on the gfx950 engine.

Mirrors the reference's ldm/models/diffusion/ddpm.py for everything the sampling harness
touches (scripts/inference_test_bench.py:403-464): DDPM.register_schedule :127-179,
q_sample :284-287, LatentDiffusion.__init__ :440-531, get_learned_conditioning :610-630,
get_input :758-834, decode_first_stage :837-901, encode_first_stage :970-1008,
encode_all_stages :1010-1033, apply_model :1060-1157, decode_sample :1420-1447,
DiffusionWrapper :1682-1722.  Training (p_losses, optimizers, EMA updates, logging images)
is out of scope.
"""
import warnings
from contextlib import contextmanager
from functools import partial

import numpy as np
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ...modules.diffusionmodules.util import Linear, extract_into_tensor, make_beta_schedule
from ...modules.distributions.distributions import DiagonalGaussianDistribution
from ...util import cat_interleave, default, instantiate_from_config, make_contiguous


class DiffusionWrapper(nn.Module):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config).eval()
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "crossattn"], "MObI uses conditioning_key: crossattn"

    def forward(self, x, t, c_concat=None, c_crossattn=None):
        if self.conditioning_key is None:
            return self.diffusion_model(x, t)
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc)


class DDPM(nn.Module):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=[], load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4,
                 linear_end=2e-2, cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0.,
                 l_simple_weight=1., conditioning_key=None, parameterization="eps", scheduler_config=None,
                 use_positional_encodings=False, learn_logvar=False, logvar_init=0., u_cond_percent=0):
        super().__init__()
        assert parameterization == "eps"
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.first_stage_key = first_stage_key
        self.image_size = image_size
        self.channels = channels
        self.u_cond_percent = u_cond_percent
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = use_ema
        if use_ema:
            raise NotImplementedError("EMA shadow weights are a training feature; MObI ships use_ema: False")
        self.v_posterior = v_posterior
        self.loss_type, self.l_simple_weight, self.original_elbo_weight = loss_type, l_simple_weight, original_elbo_weight
        if learn_logvar:
            raise NotImplementedError("learn_logvar: the engine's backward pass covers the UNet's adapter tensors, not logvar")
        self.learn_logvar = learn_logvar
        self.use_scheduler = scheduler_config is not None      # ddpm.py:97-99
        if self.use_scheduler:
            self.scheduler_config = scheduler_config
        if monitor is not None:
            self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        # (a plain tensor in the reference, indexed by a device-resident t: a buffer here so that it follows .to())
        self.register_buffer("logvar", torch.full(fill_value=logvar_init, size=(self.num_timesteps,)), persistent=False)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, only_model=load_only_unet)

    @property
    def device(self):
        return self.betas.device

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False):
        """Checkpoint ingestion with the reference's diagnostics (ddpm.py:196-212): keys under an `ignore_keys`
        prefix are dropped, the missing / unexpected key lists of the non-strict load are printed.  On top of the
        reference: a checkpoint that leaves part of the UNet or a VAE at its random initialisation samples
        garbage, so missing keys under those prefixes also raise a RuntimeWarning (not an error: initialising
        from a checkpoint without the adapter weights is a legitimate use of `strict=False`)."""
        sd = torch.load(path, map_location="cpu")
        if "state_dict" in sd:
            sd = sd["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        target = self.model if only_model else self
        missing, unexpected = target.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if len(missing) > 0:
            print(f"Missing Keys: {missing}")
        if len(unexpected) > 0:
            print(f"Unexpected Keys: {unexpected}")
        hot = ("model.diffusion_model.", "first_stage_model.", "lidar_stage_model.", "diffusion_model.")
        lost = [k for k in missing if k.startswith(hot) and not any(k.startswith(ik) for ik in ignore_keys)]
        if lost:
            warnings.warn(f"{path} leaves {len(lost)} UNet / VAE parameters at their initial values "
                          f"(first: {lost[:4]})", RuntimeWarning)
        return missing, unexpected

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", f32(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1. - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1. - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1. / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1. / alphas_cumprod - 1)))
        posterior_variance = (1 - self.v_posterior) * betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod) \
            + self.v_posterior * betas
        self.register_buffer("posterior_variance", f32(posterior_variance))
        self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(posterior_variance, 1e-20))))
        self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod)))
        self.register_buffer("posterior_mean_coef2",
                             f32((1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod)))
        # weights of the variational-bound term (ddpm.py:169-179, eps-parameterisation), in the reference's fp32 tensor ops
        lvlb = self.betas ** 2 / (2 * self.posterior_variance * f32(alphas) * (1 - self.alphas_cumprod))
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", lvlb, persistent=False)
        assert not torch.isnan(self.lvlb_weights).all()

    @contextmanager
    def ema_scope(self, context=None):
        yield None          # use_ema is False in every MObI config (mobi_nusc_512.yaml:47)

    def q_sample(self, x_start, t, noise=None):
        """sqrt(ac[t]) * x0 + sqrt(1 - ac[t]) * noise (ddpm.py:284-287): one kernel gathers the tables by the
        device-resident int64 t -- no host read-back."""
        noise = default(noise, lambda: torch.randn_like(x_start))
        return ops.q_sample(x_start.float().contiguous(), noise.float().contiguous(),
                            t.to(device=x_start.device, dtype=torch.int64).contiguous(),
                            self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    def get_input(self, batch, k):
        return make_contiguous(batch["image"]), make_contiguous(batch["lidar"])


    def get_loss(self, pred, target, mean=True):
        """ddpm.py:289-303: elementwise |target - pred| or (target - pred)^2 (a [N] .. [N, 4, h, w] fp32 tensor on the
        device), or its mean."""
        if self.loss_type == "l1":
            loss = (target - pred).abs()
            return loss.mean() if mean else loss
        if self.loss_type == "l2":
            return F.mse_loss(target, pred) if mean else F.mse_loss(target, pred, reduction="none")
        raise NotImplementedError("unknown loss type '{loss_type}'")


class LatentDiffusion(DDPM):
    def __init__(self, cond_stage_config, first_stage_config=None, lidar_stage_config=None, num_timesteps_cond=None,
                 cond_stage_key="image", cond_stage_trainable=False, concat_mode=True, cond_stage_forward=None,
                 conditioning_key=None, scale_factor=1.0, lidar_scale_factor=1.0, scale_by_std=False,
                 use_camera=True, use_lidar=False, range_object_norm=False, range_object_norm_scale=0.75,
                 range_int_norm=False, *args, **kwargs):
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        assert self.num_timesteps_cond == 1 and not scale_by_std
        self.range_object_norm, self.range_object_norm_scale = range_object_norm, range_object_norm_scale
        self.range_int_norm = range_int_norm
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        super().__init__(conditioning_key=conditioning_key, *args, **kwargs)
        self.learnable_vector = nn.Parameter(torch.randn((1, 1, 768)), requires_grad=False)
        self.bbox_uncond_vector = nn.Parameter(torch.randn((1, 1, 768)), requires_grad=False)
        self.proj_out = Linear(1024, 768)
        self.concat_mode = concat_mode
        self.cond_stage_trainable = cond_stage_trainable
        self.cond_stage_key = cond_stage_key
        self.scale_factor, self.lidar_scale_factor = scale_factor, lidar_scale_factor
        self.use_camera, self.use_lidar = use_camera, use_lidar
        if not use_camera and first_stage_config is not None:
            warnings.warn("No camera input, but first_stage_config is not None. Setting first_stage_config to None.")
            first_stage_config = None
        if not use_lidar and lidar_stage_config is not None:
            warnings.warn("No lidar input, but lidar_stage_config is not None. Setting lidar_stage_config to None.")
            lidar_stage_config = None
        self.first_stage_model = self._frozen(first_stage_config)
        self.cond_stage_model = None if cond_stage_config in (None, "__is_unconditional__") \
            else self._frozen(cond_stage_config)
        self.lidar_stage_model = self._frozen(lidar_stage_config)
        self.cond_stage_forward = cond_stage_forward
        self.clip_denoised = False
        self.restarted_from_ckpt = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)
            self.restarted_from_ckpt = True

    @staticmethod
    def _frozen(config):
        if config is None:
            return None
        model = instantiate_from_config(config).eval()
        model.requires_grad_(False)
        return model

    # ---- conditioning -----------------------------------------------------------------
    def get_learned_conditioning(self, c):
        c = self.cond_stage_model.encode(c) if hasattr(self.cond_stage_model, "encode") else self.cond_stage_model(c)
        w, b = self.proj_out.skinny()
        tok = c["ref_image_token"].float()
        n, one, d = tok.shape
        c["ref_image_token"] = ops.skinny_linear(tok.reshape(n * one, d).contiguous(), w, b).reshape(n, one, -1)
        cond = []
        if "ref_image" in self.cond_stage_key:
            cond.append(c["ref_image_token"])
        if "ref_bbox" in self.cond_stage_key:
            cond.append(c["ref_bbox_token"].float())
        return torch.cat(cond, dim=1)

    def process_conditioning(self, cond_data, force_c_encode=False):
        xc = {k: cond_data[k] for k in self.cond_stage_key}
        c = self.get_learned_conditioning(xc) if (not self.cond_stage_trainable or force_c_encode) else xc
        return c, xc

    # ---- first stage --------------------------------------------------------------------
    @torch.no_grad()
    def encode_first_stage(self, x, module_name="first_stage_model"):
        return getattr(self, module_name).encode(x)

    def get_first_stage_encoding(self, encoder_posterior, scale_factor=1):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(scale=scale_factor)
        if isinstance(encoder_posterior, torch.Tensor):
            return scale_factor * encoder_posterior
        raise NotImplementedError(type(encoder_posterior))

    def _encode_modality(self, module_name, gt, inpaint, mask, scale, noises=None):
        """One branch of encode_all_stages (ddpm.py:1013-1021): the two posterior samples and the
        nearest-resized mask are written straight into the 9-channel tensor (no torch.cat)."""
        post = self.encode_first_stage(gt, module_name)
        b, c2, h, w = post.parameters.shape
        z = torch.empty((b, c2 + 1, h, w), device=gt.device, dtype=torch.float32)
        n0, n1 = noises if noises is not None else (None, None)
        post.sample(noise=n0, scale=scale, out=z, c_off=0)
        self.encode_first_stage(inpaint, module_name).sample(noise=n1, scale=scale, out=z, c_off=c2 // 2)
        ops.nearest_resize(mask.float().contiguous(), w, w, out=z, c_off=c2)      # size = z.shape[-1] (:1020)
        return z

    def encode_all_stages(self, image_gt, image_inpaint, image_mask, range_gt, range_inpaint, range_mask,
                          noises=None):
        """`noises`: optional dict {cam_gt, cam_inpaint, lidar_gt, lidar_inpaint} for parity runs; otherwise
        drawn from the CPU generator in the reference's order (distributions.py:36)."""
        nz = noises or {}
        z_image = z_lidar = None
        if self.use_camera:
            z_image = self._encode_modality("first_stage_model", image_gt, image_inpaint, image_mask,
                                            self.scale_factor,
                                            (nz.get("cam_gt"), nz.get("cam_inpaint")) if noises else None)
        if self.use_lidar:
            z_lidar = self._encode_modality("lidar_stage_model", range_gt, range_inpaint, range_mask,
                                            self.lidar_scale_factor,
                                            (nz.get("lidar_gt"), nz.get("lidar_inpaint")) if noises else None)
        return z_image, z_lidar

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False, module_name="first_stage_model",
                           clamp=None):
        assert module_name in ["first_stage_model", "lidar_stage_model"]
        module = getattr(self, module_name)
        scale = self.scale_factor if module_name == "first_stage_model" else self.lidar_scale_factor
        z = ops.lincomb4([z.float().contiguous()], [1. / scale])
        if self.first_stage_key == "inpaint":
            z = z[:, :4, :, :].contiguous()
        return module.decode(z, clamp=clamp)

    # ---- harness entry points ---------------------------------------------------------------
    @torch.no_grad()
    def get_input(self, batch, k, force_c_encode=False, bs=None, return_vae_rec=False, noises=None):
        image_data, lidar_data = super().get_input(batch, k)

        def first(x, n):
            if isinstance(x, dict):
                return {kk: first(v, n) for kk, v in x.items()}
            return x[:n] if isinstance(x, torch.Tensor) else None

        if bs is not None:
            image_data, lidar_data = first(image_data, bs), first(lidar_data, bs)
        z_image, z_lidar = self.encode_all_stages(
            image_gt=image_data.get("GT"), image_inpaint=image_data.get("inpaint_image"),
            image_mask=image_data.get("inpaint_mask"), range_gt=lidar_data.get("range_data"),
            range_inpaint=lidar_data.get("range_data_inpaint"), range_mask=lidar_data.get("range_mask"),
            noises=noises)
        out = {"z": [], "cond": []}
        if self.use_camera:
            out["z"].append(z_image)
            c, _ = self.process_conditioning(image_data["cond"], force_c_encode=force_c_encode)
            out["cond"].append(c)
            if return_vae_rec:
                out["image_rec"] = self.decode_first_stage(z_image[:, :4, ...], clamp=(-1., 1.))
        if self.use_lidar:
            if z_lidar.shape[-1] != self.image_size:
                warnings.warn("Cropping lidar feature map to match image latent size.")
            W = z_lidar.shape[-1]
            left, right = W // 2 - self.image_size // 2, W // 2 + self.image_size // 2
            pad = (self.image_size - z_lidar.shape[-2]) // 2
            out["z"].append(F.pad(z_lidar[..., left:right], (0, 0, pad, pad), mode="constant", value=0))
            bbox = lidar_data["cond"]["ref_bbox"]                       # edited in place, as the reference does
            bbox[..., 0] = (bbox[..., 0] * W - left) / self.image_size
            bbox[..., 1] += pad / self.image_size
            c, _ = self.process_conditioning(lidar_data["cond"], force_c_encode=force_c_encode)
            out["cond"].append(c)
            out["z_lidar"] = z_lidar[:, :4, ...]
            if return_vae_rec:
                out["lidar_rec"] = self.decode_first_stage(z_lidar[:, :4, ...], module_name="lidar_stage_model",
                                                           clamp=(-1., 1.))
        out["z"] = cat_interleave(out["z"])
        if force_c_encode:
            out["cond"] = cat_interleave(out["cond"])
        else:
            out["cond"] = {kk: cat_interleave([d[kk] for d in out["cond"]]) for kk in self.cond_stage_key}
        return out

    # ---- training / validation step: the loss, and (training_step) the adapter gradients on the engine ------------------
    def shared_step(self, batch, **kwargs):
        """ddpm.py:1036-1038: batch -> (loss, loss dict)."""
        data = self.get_input(batch, self.first_stage_key)
        return self(data["z"], data["cond"])

    def _draw_step(self, x, c):
        """What ddpm.py:1040-1058 decides per call: a timestep per element, and -- with probability `u_cond_percent` -- the
        learnt unconditional tokens in place of the conditioning (the draw is kept in `self.u_cond_prop` as there)."""
        import random
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
        if self.model.conditioning_key is not None:
            assert c is not None
            if self.cond_stage_trainable and isinstance(c, dict):    # (raw conditioning inputs; tokens pass through)
                c = self.get_learned_conditioning(c)
        self.u_cond_prop = random.uniform(0, 1)
        if self.u_cond_prop < self.u_cond_percent:
            toks = [self.learnable_vector] + ([self.bbox_uncond_vector] if "ref_bbox" in self.cond_stage_key else [])
            c = torch.cat([v.repeat(x.shape[0], 1, 1) for v in toks], dim=1)
        return t, c

    def forward(self, x, c, *args, **kwargs):
        t, c = self._draw_step(x, c)
        return self.p_losses(x, c, t, *args, **kwargs)

    def _noised_input(self, x_start, t, noise):
        """The UNet's input of a training step (ddpm.py:1179-1187): the 4 latent channels noised to level t on the engine
        (`mobi_q_sample`), the inpainting channels (masked latent, mask) passed through -> (x_noisy, target = the noise)."""
        inpaint = self.first_stage_key == "inpaint"
        clean = x_start[:, :4] if inpaint else x_start
        noise = default(noise, lambda: torch.randn_like(clean))
        x_noisy = self.q_sample(x_start=clean, t=t, noise=noise)
        if inpaint:
            x_noisy = torch.cat((x_noisy, x_start[:, 4:].float()), dim=1)
        return x_noisy, noise.float()

    def _loss_terms(self, loss_simple, t):
        """loss_simple [N] -> (loss, dict) with the weights of ddpm.py:1196-1216: l_simple_weight * mean(loss_simple /
        exp(logvar_t) + logvar_t) + original_elbo_weight * mean(lvlb_weights_t * loss_simple)."""
        prefix = "train" if self.training else "val"
        logvar_t = self.logvar[t].to(self.device)
        loss_vlb = (self.lvlb_weights[t] * loss_simple).mean()
        loss = self.l_simple_weight * (loss_simple / torch.exp(logvar_t) + logvar_t).mean() + self.original_elbo_weight * loss_vlb
        return loss, {f"{prefix}/loss_simple": loss_simple.mean(), f"{prefix}/loss_vlb": loss_vlb, f"{prefix}/loss": loss}

    @torch.no_grad()
    def p_losses(self, x_start, cond, t, noise=None):
        """ddpm.py:1177-1217 (eps-parameterisation), forward only: one UNet evaluation on the engine."""
        x_noisy, target = self._noised_input(x_start, t, noise)
        model_output = self.apply_model(x_noisy, t, cond)
        return self._loss_terms(self.get_loss(model_output, target, mean=False).mean([1, 2, 3]), t)

    def validation_step(self, batch, batch_idx=0):
        """ddpm.py:372-378 without the Lightning logger: the loss dict (and its `_ema` twin: use_ema is False)."""
        _, loss_dict = self.shared_step(batch)
        with self.ema_scope():
            _, loss_dict_ema = self.shared_step(batch)
        return {**loss_dict, **{k + "_ema": v for k, v in loss_dict_ema.items()}}

    @torch.no_grad()
    def training_step(self, batch, batch_idx=0, t=None, noise=None, loss_scale=None, allreduce=True):
        """ddpm.py:356-370 of the reference + what Lightning does around it (autograd backward, DDP's gradient all-reduce,
        main.py:510): returns the loss and leaves `self.adapter_grads` = {`model.diffusion_model.<name>`: fp32 gradient} for
        every UNet tensor the reference's optimizer filter selects (ddpm.py:1616-1629: `cond_adapter*`, `cross_modal*` --
        432 tensors, 180 M parameters), computed by the engine's backward pass (mobi_amd/train.py) and summed over the ranks
        (mobi_amd.dist.allreduce_gradients).  FIRST SLICE of SURVEY 8(f) row 4: l2 loss with the default weights
        (`learn_logvar=False`, `original_elbo_weight=0`: the gradient of mean(loss_simple)); the conditioning stage's
        trainable tensors (the 3-D box embedder's four Linear layers, or `bbox_uncond_vector` on an unconditional draw) get
        theirs too when `cond_stage_trainable`; `configure_optimizers()` returns the engine's AdamW to step with them."""
        from .... import dist as mdist, engine_dtype, train
        if self.loss_type != "l2" or self.parameterization != "eps" or self.learn_logvar or self.original_elbo_weight != 0:
            raise NotImplementedError("the engine's training step covers the eps / l2 simple loss MObI trains with")
        data = self.get_input(batch, self.first_stage_key)
        x, c = data["z"], data["cond"]
        # the conditioning stage's trainable part (ddpm.py:1635-1647): the 3-D box embedder runs with a tape, so that the
        # gradient the UNet hands back for the box token reaches its four Linear layers (or `bbox_uncond_vector`)
        bbox_tape = None
        if self.cond_stage_trainable and isinstance(c, dict) and "ref_bbox" in self.cond_stage_key \
                and hasattr(self.cond_stage_model, "bbox_embedder"):
            tok = self.cond_stage_model.encode({"ref_image": c["ref_image"]})["ref_image_token"].float()
            w, b = self.proj_out.skinny()
            nn_, one, d = tok.shape
            ref_tok = ops.skinny_linear(tok.reshape(nn_ * one, d).contiguous(), w, b).reshape(nn_, one, -1)
            box_tok, bbox_tape = train.bbox_embedder_forward(self.cond_stage_model.bbox_embedder, c["ref_bbox"])
            c = torch.cat([ref_tok, box_tok.float()], dim=1)
        cond_was_given = c
        t_draw, c = self._draw_step(x, c)
        uncond = c is not cond_was_given and self.u_cond_prop < self.u_cond_percent
        t = t_draw if t is None else t
        x_noisy, target = self._noised_input(x, t, noise)
        if loss_scale is None:
            # fp16 gradients underflow without it; bf16 has the range.  The gradient that enters the network is
            # 2 (eps - target) / numel, so the scale follows numel (a power of two near numel / 4: the entering gradient is
            # O(1) at every batch size) -- measured on the full-width network at 64 x 64, one pair: all 432 gradients 1.6e-2
            # off with a fixed 256, 1.7e-3 with 8,192 (tests/test_gpu_backward.py, full width)
            loss_scale = 2.0 ** round(math.log2(max(4, target.numel()) / 4)) if engine_dtype() == torch.float16 else 1.0
        logvar_t = self.logvar[t].to(self.device)                      # (zeros unless a checkpoint says otherwise)
        if bool((logvar_t != 0).any()):
            raise NotImplementedError("per-timestep logvar weights in the backward pass")
        mse, grads = train.loss_and_gradients(self.model.diffusion_model, x_noisy, t, c, target, loss_scale=loss_scale)
        if self.l_simple_weight != 1.0:
            grads = {k: ops.lincomb4([g.contiguous()], [float(self.l_simple_weight)]) for k, g in grads.items()}
        dctx = grads.pop("__dcontext__", None)                         # fp32 [N, 2, ctx_dim]
        named = {"model.diffusion_model." + k: v for k, v in grads.items()}
        if self.cond_stage_trainable and dctx is not None and "ref_bbox" in self.cond_stage_key:
            dbox = dctx[:, 1].contiguous()
            if uncond:                                                 # the learnt unconditional box token stood in for every element
                ones = torch.ones((1, dbox.shape[0]), device=dbox.device, dtype=torch.float32)
                named["bbox_uncond_vector"] = ops.linear_f32(dbox.t().contiguous(), ones).reshape(1, 1, -1)
            elif bbox_tape is not None:
                eg = train.bbox_embedder_backward(self.cond_stage_model.bbox_embedder, bbox_tape, dbox.unsqueeze(1))
                named.update({"cond_stage_model.bbox_embedder." + k: v for k, v in eg.items()})
        # every rank brings the SAME tensors to the collective, whatever ITS draw was: an unconditional draw has a gradient for
        # `bbox_uncond_vector` and none for the box embedder, a conditional one the other way round, and the draws are per
        # process (the buckets of allreduce_gradients are cut by name and size) -- the branch not taken contributes zeros, which
        # is also what DDP sums for a parameter that a rank's graph did not reach
        # (a tensor NO rank has a gradient for stays absent, as under DDP, where its `.grad` stays None and AdamW skips it;
        #  one process alone completes nothing)
        if allreduce and mdist.world()[1] > 1:
            mdist.allreduce_gradients(self._complete_cond_stage_grads(named, x.device, across_ranks=True))
        self.adapter_grads = named
        return self.l_simple_weight * mse

    def _cond_stage_trainables(self):
        """{name: parameter} of the conditioning stage's trainable tensors (ddpm.py:1635-1647): the 3-D box embedder's Linear
        layers and `bbox_uncond_vector` -- the same set `configure_optimizers` hands to AdamW."""
        out = {}
        if self.cond_stage_trainable and "ref_bbox" in self.cond_stage_key and hasattr(self.cond_stage_model, "bbox_embedder"):
            out.update({"cond_stage_model.bbox_embedder." + n: p for n, p in self.cond_stage_model.bbox_embedder.named_parameters()
                        if "class_embedder" not in n})
            out["bbox_uncond_vector"] = self.bbox_uncond_vector
        return out

    def _complete_cond_stage_grads(self, named, device, across_ranks=False):
        """Zeros for every conditioning-stage tensor `named` has no gradient for (the branch this rank's draw did not take).
        across_ranks: only the tensors SOME rank has a gradient for (one MAX all-reduce of a presence mask, in the fixed
        order of `_cond_stage_trainables`) -- every rank ends up with the same key set."""
        tr = self._cond_stage_trainables()
        keys = list(tr)
        present = torch.tensor([1 if k in named else 0 for k in keys], dtype=torch.int32, device=device)
        if across_ranks and keys:
            import torch.distributed as tdist
            tdist.all_reduce(present, op=tdist.ReduceOp.MAX)
        elif keys:
            present.fill_(1)
        for k, have in zip(keys, present.tolist()):
            if have and k not in named:
                named[k] = torch.zeros(tuple(tr[k].shape), device=device, dtype=torch.float32)
        return named

    def configure_optimizers(self):
        """ddpm.py:1616-1669 of the reference: AdamW (lr = `self.learning_rate`) over the UNet tensors whose names contain
        `cond_adapter`, `lidar` or `cross_modal`, plus -- with a trainable conditioning stage -- the box embedder and
        `bbox_uncond_vector`; here the engine's AdamW (mobi_amd.train.AdamW, `mobi_adamw_step`), stepped with
        `self.adapter_grads` after `training_step`.  With a `scheduler_config` (every MObI config: LambdaLinearScheduler, 200 warm-up
        steps) the return value has the reference's Lightning form `([opt], [{scheduler, interval: "step", frequency: 1}])`, the
        scheduler a `train.LambdaLR` over the config's schedule object (ddpm.py:1651-1668): call `scheduler.step()` after
        every optimizer step."""
        from .... import train
        params = {"model.diffusion_model." + n: p for n, p in self.model.diffusion_model.named_parameters()
                  if any(m in n for m in train.TRAINABLE_MARKERS)}
        params.update(LatentDiffusion._cond_stage_trainables(self))
        opt = train.AdamW(params, lr=getattr(self, "learning_rate", 1e-4))
        if self.use_scheduler:
            assert "target" in self.scheduler_config
            schedule = instantiate_from_config(self.scheduler_config)
            return [opt], [{"scheduler": train.LambdaLR(opt, lr_lambda=schedule.schedule), "interval": "step", "frequency": 1}]
        return opt

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """x_noisy: fp32 [N, 9, h, w] or the un-concatenated list [x, inpaint_image, inpaint_mask]."""
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            cond = {"c_crossattn": cond}
        return self.model(x_noisy, t, **cond)

    @torch.no_grad()
    def decode_sample(self, sample, z_lidar=None):
        h_camera = h_lidar = None
        if self.use_camera and self.use_lidar:
            h_camera = sample[::2]
            lid = sample[1::2]
            bottom = (lid.shape[-2] - z_lidar.shape[-2]) // 2
            h_lidar = lid[:, :, bottom:bottom + z_lidar.shape[-2], :]
            if self.image_size != z_lidar.shape[-1]:
                c = z_lidar.shape[-1] // 2
                z_lidar[..., c - self.image_size // 2: c + self.image_size // 2] = h_lidar
                h_lidar = z_lidar
        elif self.use_camera:
            h_camera = sample
        else:
            bottom = (sample[1::2].shape[-2] - z_lidar.shape[-2]) // 2
            h_lidar = sample[:, :, bottom:bottom + z_lidar.shape[-2], :]
            if self.image_size != z_lidar.shape[-1]:
                c = z_lidar.shape[-1] // 2
                z_lidar[..., c - self.image_size // 2: c + self.image_size // 2] = h_lidar
                h_lidar = z_lidar
        return h_camera, h_lidar

    @torch.no_grad()
    def log_data(self, batch, data, h_camera, h_lidar, log_metrics=True, return_sample=False, split="train"):
        """ddpm.py:1471-1612 of the reference with the arithmetic on the device:
          decode + clamp (:1475-1476, :1503-1504); the uint8 collages `image_preds` / `image_preds_no_box` /
          `image_input-rec` (:1478-1497) and `range_depth_pred` / `range_int_pred` (:1521-1522); the range-view
          de-normalisation (:1527-1543, `mobi_range_denorm`); the per-sample lidar error scores (:1545-1590) as ONE
          device table per (pred, gt) pair (`mobi_lidar_metrics`) and one read-back for all of them instead of ~100
          `.item()` syncs; the point-cloud pictures (:1600-1612).  As in the reference, `range_sample_depth` is the
          DE-NORMALISED depth (the logged tensor is overwritten in place, :1533-1537) while `range_sample_int` stays
          the clamped raw intensity (:1541 makes a new tensor).  Parts whose inputs the batch does not carry are
          skipped (bench.py passes only what a throughput run needs).  Box outlines need cv2; without it
          `image_preds` equals `image_preds_no_box`."""
        from ...data import utils as du
        log, lidar_metrics = dict(), None
        has = lambda d, *ks: isinstance(d, dict) and all(k in d for k in ks)
        if self.use_camera:
            image_sample = self.decode_first_stage(h_camera, clamp=(-1., 1.))
            img = batch.get("image") if isinstance(batch, dict) else None
            if has(img, "GT", "inpaint_image", "cond") and has(data, "image_rec"):
                dev = image_sample.device
                u8 = lambda x, clip=False: ((du.un_norm_clip(x.to(dev).float()) if clip else du.un_norm(x.to(dev).float()))
                                            * 255).to(torch.int32).bitwise_and(255).to(torch.uint8)
                sample, inp, inpaint = u8(image_sample), u8(img["GT"]), u8(img["inpaint_image"])
                ref, rec = u8(img["cond"]["ref_image"], clip=True), u8(data["image_rec"])
                log["image_preds_no_box"] = torch.cat([inp, inpaint, ref, sample], dim=-2)
                boxed = du.draw_boxes_if_possible([inp, inpaint, sample, rec], img["cond"].get("ref_bbox"))
                log["image_preds"] = torch.cat([boxed[0], boxed[1], ref, boxed[2]], dim=-2)
                log["image_input-rec"] = torch.cat([boxed[0], boxed[3]], dim=-2)
            if return_sample:
                log["image_sample"] = image_sample
        if self.use_lidar:
            lidar_sample = self.decode_first_stage(h_lidar, module_name="lidar_stage_model", clamp=(-1., 1.))
            if return_sample:                      # (not a key of the reference's dict: a 2-channel tensor would break
                log["lidar_sample"] = lidar_sample  # consumers that turn every entry into a picture, main.py:341-387)
            lid = batch.get("lidar") if isinstance(batch, dict) else None
            if lid is None:
                return log, lidar_metrics
            dev = lidar_sample.device
            smp = lidar_sample.float().contiguous()
            full = has(lid, "range_data", "range_data_inpaint", "range_mask", "range_instance_mask") and has(data, "lidar_rec")
            if full:
                rd, rdi = lid["range_data"].to(dev).float(), lid["range_data_inpaint"].to(dev).float()
                rec = data["lidar_rec"].to(dev).float().contiguous()
                inst = lid["range_instance_mask"].to(dev).float()
                box = 1 - lid["range_mask"].to(dev).float()[:, [0]]
                log["range_depth_pred"] = torch.cat([rd[:, [0]], rdi[:, [0]], inst, smp[:, [0]], rec[:, [0]]], dim=-2)
                log["range_int_pred"] = torch.cat([rd[:, [1]], rdi[:, [1]], inst, smp[:, [1]], rec[:, [1]]], dim=-2)
            has_range = self.range_object_norm and has(lid, "min_depth_obj", "max_depth_obj")
            lo = hi = None
            if has_range:
                lo = torch.as_tensor(lid["min_depth_obj"], dtype=torch.float32, device=dev).reshape(-1).contiguous()
                hi = torch.as_tensor(lid["max_depth_obj"], dtype=torch.float32, device=dev).reshape(-1).contiguous()
            denorm = lambda t: ops.range_denorm(t.contiguous(), lo, hi, alpha=self.range_object_norm_scale,
                                                object_norm=bool(has_range), int_norm=bool(self.range_int_norm))
            depth, inten = denorm(smp)
            if return_sample:
                log["range_sample_depth"] = depth
                log["range_sample_int"] = smp[:, [1]]
                log["range_sample_int_denorm"] = inten    # (opt-in like lidar_sample; the reference keeps it local)
                if "range_mask" in lid:
                    log["range_bbox_mask"] = 1 - lid["range_mask"].to(dev)[:, [0]]
            if full and "width_crop" in lid:
                in_depth, in_int = denorm(rd.contiguous())
                rec_depth, rec_int = denorm(rec)
                pairs = {"pred_depth": (depth, in_depth), "rec_depth": (rec_depth, in_depth),
                         "pred_int": (inten, in_int), "rec_int": (rec_int, in_int)}
                wc = lid["width_crop"]
                table = torch.stack([ops.lidar_metrics(p[:, 0], g[:, 0], inst[:, 0], box[:, 0], wc)
                                     for p, g in pairs.values()]).cpu().numpy()        # [4, B, 2, 3]: ONE read-back
                lidar_metrics = {}
                for pi, name in enumerate(pairs):
                    for si, score in ((0, "mse"), (1, "median_error")):
                        obj = table[pi, :, 0, si]
                        obj = obj[~np.isnan(obj)]                                       # `del object_scores[-1]` on NaN
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore", RuntimeWarning)              # np.mean([]) = nan, as there
                            lidar_metrics[f"{score}/object_{name}"] = np.mean(obj.astype(np.float64))
                            lidar_metrics[f"{score}/mask_{name}"] = np.mean(table[pi, :, 1, si].astype(np.float64))
                lidar_metrics = {f"{split}/{k}": v * ((54 - 1.4) / 2) if "depth" in k else v * 128
                                 for k, v in lidar_metrics.items()}
                if has(batch, "bbox_3d") and has(lid, "range_depth_orig", "range_shift_left", "range_pitch", "range_yaw"):
                    vis = du.get_lidar_vis(sample=depth, input=in_depth, rec=rec_depth, bboxes=batch["bbox_3d"],
                                           range_depth_orig=lid["range_depth_orig"], range_shift_left=lid["range_shift_left"],
                                           range_pitch=lid["range_pitch"], range_yaw=lid["range_yaw"], width_crop=wc)
                    log["lidar_input-pred-rec"] = torch.cat([vis[1], vis[0], vis[2]], dim=-2)
        return log, lidar_metrics
