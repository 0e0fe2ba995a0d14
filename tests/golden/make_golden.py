#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Run in the build container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/make_golden.py

Each fixture holds inputs that are cheap to store plus the reference's outputs;
parameters are NOT stored -- they are regenerated on both sides from
`oracle.weights.synth_param(key, shape, seed)` (counter-based, version
independent).  The reference code is imported from where it lies; nothing of it
is copied into this repo.

Stubs: the reference imports packages this image lacks (pytorch_lightning,
omegaconf, torchvision, cv2, taming, ...) at module top level; they are replaced
by minimal stand-ins *for the import only* -- no arithmetic of the hot path
lives in them.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MOBI_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from oracle import weights as W                      # noqa: E402


def install_stubs():
    import torch.nn as nn
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed: its availability probe)
    from transformers import CLIPVisionModel  # noqa: F401

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl = mod("pytorch_lightning", LightningModule=LightningModule)
    mod("pytorch_lightning.utilities")
    mod("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]

    class ListConfig(list):
        pass

    mod("omegaconf", ListConfig=ListConfig)
    mod("omegaconf.listconfig", ListConfig=ListConfig)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms = mod("torchvision.transforms", Resize=object)
    mod("cv2")
    mod("taming")
    mod("taming.modules")
    mod("taming.modules.vqvae")
    mod("taming.modules.vqvae.quantize", VectorQuantizer2=object)
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.cm", "pandas_stub"):
        pass


def import_reference():
    install_stubs()
    # the reference's `ldm` directory has no __init__.py (a namespace package), so ANY regular package called `ldm`
    # further down sys.path would win over it -- and the repo root holds one (the alias of mobi_amd.ldm).  Take the repo
    # root off the path for the import (everything this script needs from the repo is imported already).
    for name in [n for n in sys.modules if n == "ldm" or n.startswith("ldm.")]:
        del sys.modules[name]
    sys.meta_path[:] = [f for f in sys.meta_path if type(f).__name__ != "_AliasFinder"]
    sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != REPO]
    sys.path.insert(0, REF)
    import ldm.modules.diffusionmodules.openaimodel as om
    import ldm.modules.diffusionmodules.model as vm
    import ldm.modules.diffusionmodules.util as ut
    import ldm.modules.attention as at
    import ldm.models.diffusion.ddim as ddim
    import ldm.models.diffusion.plms as plms
    import ldm.modules.distributions.distributions as dist
    import ldm.util as lu
    try:
        import ldm.models.diffusion.ddpm as ddpm
    except Exception as e:                                   # pragma: no cover
        print("ddpm import failed:", repr(e))
        ddpm = None
    assert os.path.abspath(om.__file__).startswith(os.path.abspath(REF)), om.__file__
    return types.SimpleNamespace(om=om, vm=vm, ut=ut, at=at, ddim=ddim, plms=plms, dist=dist, lu=lu, ddpm=ddpm)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------
# shared case definitions (also imported by the tests)
# ---------------------------------------------------------------------------
UNET_CASES = {
    # name: (UNetModel kwargs, batch, latent side)
    "unet_mc32_mm": (dict(image_size=16, in_channels=9, out_channels=4, model_channels=32,
                          attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
                          num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                          legacy=False, bbox_cond=True, use_camera=True, use_lidar=True), 4, 16),
    "unet_mc64_mm": (dict(image_size=8, in_channels=9, out_channels=4, model_channels=64,
                          attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
                          num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                          legacy=False, bbox_cond=True, use_camera=True, use_lidar=True), 2, 8),
    "unet_mc32_cam": (dict(image_size=16, in_channels=9, out_channels=4, model_channels=32,
                           attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
                           num_heads=8, use_spatial_transformer=True, transformer_depth=1, context_dim=768,
                           legacy=False, bbox_cond=False, use_camera=True, use_lidar=False), 3, 16),
}

VAE_CASES = {
    # name: (ddconfig, embed_dim, batch, resolution)
    "vae_cam32": (dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
                       ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0), 4, 2, 64),
    "vae_lidar32": (dict(double_z=True, z_channels=4, resolution=64, in_channels=2, out_ch=2, ch=32,
                         ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], lidar_adapter=True,
                         dropout=0.0), 4, 2, 64),
}


def unet_inputs(name, batch, side, in_ch=9, ctx_dim=768):
    x = W.synth_input(name + ".x", (batch, in_ch, side, side))
    ctx = W.synth_input(name + ".ctx", (batch, 2, ctx_dim))
    t = torch.tensor([(37 + 211 * i) % 1000 for i in range(batch)], dtype=torch.long)
    return x, t, ctx


def main():
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    R = import_reference()
    om, vm, ut, at = R.om, R.vm, R.ut, R.at

    # ---- 1. schedules and DDIM tables (bit-exact part) -----------------------
    betas = ut.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    tabs = {"betas_f64": betas}
    for S in (10, 50, 250, 30):
        tabs[f"ddim_timesteps_S{S}"] = ut.make_ddim_timesteps("uniform", S, 1000, verbose=False)

    class FakeDDPM(torch.nn.Module):
        v_posterior = 0.0
        parameterization = "eps"

        @property
        def device(self):
            return torch.device("cpu")

    fake = FakeDDPM()
    R.ddpm.DDPM.register_schedule(fake, beta_schedule="linear", timesteps=1000,
                                  linear_start=0.00085, linear_end=0.012)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        tabs["ddpm_" + k] = getattr(fake, k)
    fake.q_sample = lambda x0, t, noise=None: R.ddpm.DDPM.q_sample(fake, x0, t, noise)

    R.ddim.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)   # off "cuda" (ddim.py:19-23)
    R.plms.PLMSSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    for S in (10, 50):
        for eta in (0.0, 1.0):
            smp = R.ddim.DDIMSampler(fake)
            smp.make_schedule(S, ddim_eta=eta, verbose=False)
            tag = f"S{S}_eta{int(eta)}"
            for k in ("ddim_sigmas", "ddim_alphas", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas"):
                v = getattr(smp, k)
                tabs[f"{k}_{tag}"] = np.asarray(v.numpy() if isinstance(v, torch.Tensor) else v)
    save("schedule_tables", **tabs)

    # ---- 2. timestep embedding + per-operator goldens -------------------------
    ops = {}
    t = torch.tensor([1, 21, 500, 981, 999], dtype=torch.long)
    ops["temb_t"] = t
    ops["temb_320"] = ut.timestep_embedding(t, 320)
    ops["temb_32"] = ut.timestep_embedding(t, 32)

    emb = W.synth_input("ops.emb", (4, 128))
    for tag, cin, cout in (("res64", 64, 64), ("res96to64", 96, 64)):
        m = om.ResBlock(cin, 128, 0.0, out_channels=cout).eval()
        W.fill_module_(m, seed=1, prefix=tag + ".")
        x = W.synth_input(f"ops.{tag}.x", (4, cin, 8, 8))
        ops[tag + "_y"] = m(x, emb)
    m = om.Downsample(64, True, out_channels=64).eval()
    W.fill_module_(m, seed=1, prefix="down64.")
    ops["down64_y"] = m(W.synth_input("ops.down64.x", (2, 64, 8, 8)))
    m = om.Upsample(64, True, out_channels=64).eval()
    W.fill_module_(m, seed=1, prefix="up64.")
    ops["up64_y"] = m(W.synth_input("ops.up64.x", (2, 64, 4, 4)))

    m = at.SpatialTransformer(64, 8, 8, depth=1, context_dim=768, bbox_cond=True, multimodal=True).eval()
    W.fill_module_(m, seed=1, prefix="st64.")
    ops["st64_y"] = m(W.synth_input("ops.st64.x", (4, 64, 8, 8)), W.synth_input("ops.st64.ctx", (4, 2, 768)))

    for tk in (1, 2, 16):
        m = at.CrossAttention(64, context_dim=48, heads=8, dim_head=8).eval()
        W.fill_module_(m, seed=1, prefix=f"xattn{tk}.")
        ops[f"xattn{tk}_y"] = m(W.synth_input(f"ops.xattn{tk}.x", (2, 16, 64)),
                                context=W.synth_input(f"ops.xattn{tk}.ctx", (2, tk, 48)))
    m = at.FeedForward(64, glu=True).eval()
    W.fill_module_(m, seed=1, prefix="ff64.")
    ops["ff64_y"] = m(W.synth_input("ops.ff64.x", (2, 16, 64)))

    for tag, ks, pad in (("vres3", 3, 1), ("vres15", (1, 5), (0, 2))):
        m = vm.ResnetBlock(in_channels=32, out_channels=64, temb_channels=0, dropout=0.0,
                           kernel_size=ks, padding=pad).eval()
        W.fill_module_(m, seed=1, prefix=tag + ".")
        ops[tag + "_y"] = m(W.synth_input(f"ops.{tag}.x", (2, 32, 8, 8)), None)
    m = vm.AttnBlock(64).eval()
    W.fill_module_(m, seed=1, prefix="vattn.")
    ops["vattn_y"] = m(W.synth_input("ops.vattn.x", (2, 64, 8, 8)))
    m = vm.Downsample(32, True).eval()
    W.fill_module_(m, seed=1, prefix="vdown.")
    ops["vdown_y"] = m(W.synth_input("ops.vdown.x", (2, 32, 8, 8)))
    mom = W.synth_input("ops.dg.moments", (2, 8, 4, 4)) * 3.0
    noise = W.synth_input("ops.dg.noise", (2, 4, 4, 4))
    dg = R.dist.DiagonalGaussianDistribution(mom)
    ops["dg_sample"] = dg.mean + dg.std * noise            # distributions.py:36 with the noise supplied
    ops["dg_mode"] = dg.mode()
    save("ops", **ops)

    # ---- 3. whole UNet at reduced width ----------------------------------------
    unets = {}
    for name, (kw, batch, side) in UNET_CASES.items():
        net = om.UNetModel(**kw).eval()
        W.fill_module_(net, seed=7)
        x, t, ctx = unet_inputs(name, batch, side)
        y = net(x, t, context=ctx)
        unets[name] = net
        assert float(y.abs().max()) > 0
        save(name, x=x, t=t, ctx=ctx, y=y, n_params=np.int64(sum(p.numel() for p in net.parameters())))

    # ---- 4. VAE at reduced width -------------------------------------------------
    from ldm.models.autoencoder import AutoencoderKL
    vaes = {}
    for name, (dd, ed, batch, res) in VAE_CASES.items():
        vae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=ed).eval()
        W.fill_module_(vae, seed=11)
        vaes[name] = vae
        x = W.synth_input(name + ".x", (batch, dd["in_channels"], res, res), kind="uniform")
        post = vae.encode(x)
        noise = W.synth_input(name + ".noise", tuple(post.mean.shape))
        z = post.mean + post.std * noise
        rec = vae.decode(z)
        save(name, x=x, moments=post.parameters, noise=noise, z=z, rec=rec)

    # ---- 5. sampler trajectories -----------------------------------------------------
    net = unets["unet_mc32_mm"]
    fake.apply_model = lambda x, t, c: net(x, t, context=c)
    fake.betas = fake.betas
    b, side = 4, 16
    x_T = W.synth_input("smp.x_T", (b, 4, side, side))
    inp = W.synth_input("smp.inpaint", (b, 4, side, side))
    msk = (W.synth_input("smp.mask", (b, 1, side, side)) > 0).float()
    cond = W.synth_input("smp.cond", (b, 2, 768))
    uc = W.synth_input("smp.uc", (1, 2, 768)).repeat(b, 1, 1)
    smp_out = dict(x_T=x_T, inpaint=inp, mask=msk, cond=cond, uc=uc)
    for scale in (1.0, 5.0):
        tag = f"cfg{int(scale)}"
        s = R.ddim.DDIMSampler(fake)
        samples, inter = s.sample(S=10, batch_size=b, shape=[4, side, side], conditioning=cond, verbose=False,
                                  eta=0.0, x_T=x_T, unconditional_guidance_scale=scale,
                                  unconditional_conditioning=uc, log_every_t=3,
                                  test_model_kwargs={"inpaint_image": inp, "inpaint_mask": msk})
        smp_out[f"ddim_{tag}_samples"] = samples
        smp_out[f"ddim_{tag}_pred_x0"] = torch.stack(inter["pred_x0"][1:])
        smp_out[f"ddim_{tag}_x_inter"] = torch.stack(inter["x_inter"][1:])
        p = R.plms.PLMSSampler(fake)
        samples, inter = p.sample(S=10, batch_size=b, shape=[4, side, side], conditioning=cond, verbose=False,
                                  eta=0.0, x_T=x_T, unconditional_guidance_scale=scale,
                                  unconditional_conditioning=uc, log_every_t=3,
                                  inpaint_image=inp, inpaint_mask=msk)
        smp_out[f"plms_{tag}_samples"] = samples
        smp_out[f"plms_{tag}_pred_x0"] = torch.stack(inter["pred_x0"][1:])

    # mask-mode compositing (ddim.py:145-148): noise for q_sample and the per-step
    # (sigma-weighted, eta=1) noise come from torch's CPU generator -> replay the draws.
    x0 = W.synth_input("smp.x0", (b, 4, side, side))
    cmask = (W.synth_input("smp.cmask", (b, 1, side, side)) > 0).float()
    s = R.ddim.DDIMSampler(fake)
    torch.manual_seed(1234)
    samples, inter = s.sample(S=10, batch_size=b, shape=[4, side, side], conditioning=cond, verbose=False,
                              eta=1.0, x_T=x_T, mask=cmask, x0=x0, log_every_t=3,
                              test_model_kwargs={"inpaint_image": inp, "inpaint_mask": msk})
    torch.manual_seed(1234)
    mask_noise, step_noise = [], []
    for _ in range(10):
        mask_noise.append(torch.randn_like(x0))            # q_sample's randn_like (ddpm.py:285)
        step_noise.append(torch.randn((b, 4, side, side)))  # noise_like (ddim.py:209)
    smp_out.update(x0=x0, cmask=cmask, mask_noise=torch.stack(mask_noise), step_noise=torch.stack(step_noise),
                   ddim_mask_samples=samples, ddim_mask_x_inter=torch.stack(inter["x_inter"][1:]))
    save("sampler", **smp_out)

    # ---- 6. LatentDiffusion plumbing (encode_all_stages / lidar align / decode_sample) --
    LD = R.ddpm.LatentDiffusion
    cam, lid = vaes["vae_cam32"], vaes["vae_lidar32"]

    class FakeLD:
        use_camera = True
        use_lidar = True
        scale_factor = 0.18215
        lidar_scale_factor = 0.18215
        first_stage_key = "inpaint"
        image_size = 8
        first_stage_model = cam
        lidar_stage_model = lid
        encode_first_stage = lambda self, x, module_name="first_stage_model": getattr(self, module_name).encode(x)
        get_first_stage_encoding = lambda self, p, scale_factor=1: LD.get_first_stage_encoding(self, p, scale_factor)

    f = FakeLD()
    B = 2
    img = W.synth_input("pl.img", (B, 3, 64, 64), kind="uniform")
    # encode_all_stages resizes the mask to a SQUARE (ddpm.py:1030), so range views are square
    rng = W.synth_input("pl.range", (B, 2, 64, 64), kind="uniform")
    imask = torch.ones(B, 1, 64, 64)
    imask[:, :, 16:48, 16:48] = 0
    rmask = torch.ones(B, 1, 64, 64)
    rmask[:, :, 8:40, 20:52] = 0
    torch.manual_seed(99)
    z_image, z_lidar = LD.encode_all_stages(f, img, img * imask, imask, rng, rng * rmask, rmask)
    torch.manual_seed(99)
    noises = [torch.randn(B, 4, 8, 8), torch.randn(B, 4, 8, 8), torch.randn(B, 4, 8, 8), torch.randn(B, 4, 8, 8)]
    sample = W.synth_input("pl.sample", (2 * B, 4, 8, 8))
    h_cam, h_lid = LD.decode_sample(f, sample, z_lidar[:, :4].clone())
    save("plumbing", img=img, rng=rng, imask=imask, rmask=rmask,
         n_cam_gt=noises[0], n_cam_inp=noises[1], n_lid_gt=noises[2], n_lid_inp=noises[3],
         z_image=z_image, z_lidar=z_lidar, sample=sample, h_cam=h_cam, h_lid=h_lid,
         cat_interleave=R.lu.cat_interleave([torch.arange(6.).reshape(3, 2), -torch.arange(6.).reshape(3, 2)]))

    # ---- 7. conditioning producer: CLIP tower (tiny depth, real width) + mapper + bbox MLP ----
    import transformers
    import ldm.modules.encoders.modules as em
    clip_cfg = dict(COND_CLIP_CFG)
    orig = transformers.CLIPVisionModel.from_pretrained
    em.CLIPVisionModel.from_pretrained = classmethod(
        lambda cls, *a, **k: transformers.CLIPVisionModel(transformers.CLIPVisionConfig(**clip_cfg)))
    try:
        enc = em.FrozenCLIPImageEmbedder(conditions=["ref_image", "ref_bbox"]).eval()
    finally:
        em.CLIPVisionModel.from_pretrained = orig
    W.fill_module_(enc, seed=13)
    ref_image = W.synth_input("cond.ref_image", (2, 3, 28, 28))
    ref_bbox = W.synth_input("cond.ref_bbox", (2, 8, 3), kind="uniform") * 0.5 + 0.5
    out = enc.encode({"ref_image": ref_image, "ref_bbox": ref_bbox})
    save("cond_producer", ref_image=ref_image, ref_bbox=ref_bbox, ref_image_token=out["ref_image_token"],
         ref_bbox_token=out["ref_bbox_token"])


# real ViT-L width (the reference hard-codes 1024 for the mapper), reduced depth / resolution
COND_CLIP_CFG = dict(hidden_size=1024, intermediate_size=256, num_hidden_layers=1, num_attention_heads=16,
                     image_size=28, patch_size=14, projection_dim=64, hidden_act="quick_gelu")


if __name__ == "__main__":
    main()
