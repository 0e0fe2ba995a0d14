"""GPU parity at PRODUCTION shapes and at the north star's tolerance (BASELINE.json: "outputs within 1e-3 rel-L2 of
reference ... within a stated fp16 tolerance"), inside the driver-run `-m gpu` suite:

  * mobi_nusc_512's full-width UNet (1.04 B parameters) at the benched shape -- latent 64x64, UNet batch 16 = 8
    camera/lidar pairs -- against the CPU oracle evaluated pair by pair (objects are independent; a pair is the unit of
    coupling, ldm/modules/attention.py:245-263 of the reference); fp16 and bf16 storage.  This is the only place the
    ping-pong igemm at m = 65,536 / K = 2,880 ... 23,040, its 256-block persistent walk and dh = 40 attention inside
    the model are compared with something other than themselves.
  * BASELINE config 2 (mobi_nusc_256, batch 4): full width at 32x32, UNet batch 8.
  * trajectories on the reduced net: DDIM-50 (fp16 <= 1e-3 asserted: the north-star figure) and the DDIM-250 schedule
    of config 5 ([1, 5, ..., 997], fp16).
  * the full-width VAEs (ch = 128: mid.attn_1 at T = 1,024 / c = 512, the 1x5 lidar adapter at 128 channels) on one
    256x256 camera image and one 256x256 range view.
  * the ping-pong igemm race screen (tools/race_screen.py).

Tolerances = at most 2x the values measured on the MI355X (profiles/r05_error_table.txt, written by the suite itself under
MOBI_RECORD_ERRORS), stated per storage type.
"""
import functools
import os

import numpy as np
import pytest
import torch

from oracle import sampler as osampler, unet as ounet, vae as ovae, weights as W
from tests.golden_cases import check, record, rel_l2
from tests.test_gpu_models import _unet, _vae

pytestmark = pytest.mark.gpu

DT = [torch.float16, torch.bfloat16]
# rel-L2 vs the fp32 CPU oracle, <= 2x the values measured on the MI355X (profiles/r05_error_table.txt; first measured in round 2): full
# width forward fp16 1.36e-3 / bf16 1.07e-2 (worst pair 2.09e-3 / 1.36e-2); VAE 1.76e-3 / 1.43e-2; DDIM-50 3.9e-4 / 2.97e-3
# without guidance (the 1e-3 of fp16 is the north star's own number), 1.03e-3 / 9.2e-3 with scale 5; DDIM-250 3.5e-4
TOL_FULL = {torch.float16: 2.6e-3, torch.bfloat16: 1.9e-2}          # one full-width UNet forward
TOL_VAE = {torch.float16: 3e-3, torch.bfloat16: 2.4e-2}
TOL_DDIM50 = {(torch.float16, 1.0): 1e-3, (torch.float16, 5.0): 2.1e-3,
              (torch.bfloat16, 1.0): 6e-3, (torch.bfloat16, 5.0): 1.8e-2}
TOL_DDIM250 = {torch.float16: 7e-4}


def _set(dtype):
    import mobi_amd
    mobi_amd.set_engine_dtype(dtype)


def _threads():
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))


@functools.lru_cache(maxsize=None)
def _full_width_case(side, n):
    """(x, ctx, t, oracle output) of the full-width UNet, oracle evaluated pair by pair: read from
    tests/golden/oracle_outputs.npz (tests/oracle_cases.py; tests/test_oracle_outputs_cpu.py re-derives a sample of it
    live), computed here when the file lacks it."""
    from tests import oracle_cases as oc
    x, ctx, t = oc.prod_inputs(side, n)
    return x, ctx, t, oc.prod_forward(side, n)


@functools.lru_cache(maxsize=None)
def _full_width_net():
    cfg = ounet.UNetConfig()
    net = _unet(cfg, 64)
    net.load_state_dict(W.synth_state_dict(ounet.unet_param_shapes(cfg), 13))
    return net.cuda()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("side,n", [(64, 16), (32, 8)], ids=["nusc512_b16", "nusc256_b8"])
def test_full_width_forward_vs_oracle(dtype, side, n):
    _set(dtype)
    x, ctx, t, ref = _full_width_case(side, n)
    net = _full_width_net()
    y = net(x.cuda(), t.cuda(), context=ctx.cuda())
    assert y.shape == ref.shape and bool(torch.isfinite(y).all())
    check(rel_l2(y.cpu(), ref), TOL_FULL[dtype], f"unet_full_width_{side}x{side}_b{n}_{dtype}")
    # every pair on its own is within the same bound (an error concentrated in one image must not hide in the norm)
    worst = max(rel_l2(y[i:i + 2].cpu(), ref[i:i + 2]) for i in range(0, n, 2))
    check(worst, TOL_FULL[dtype] * 1.4, f"unet_full_width_{side}x{side}_b{n}_{dtype}_worst_pair")
    # split-K slabs summed by the consuming GroupNorm (ops.Deferred, the default) against every split launch finishing itself
    # with its reduce launch: the same arithmetic in the same order, so the same bits
    from mobi_amd import ops
    assert ops.DEFER_SPLIT
    ops.DEFER_SPLIT = False
    try:
        y2 = net(x.cuda(), t.cuda(), context=ctx.cuda())
    finally:
        ops.DEFER_SPLIT = True
    assert torch.equal(y, y2)


def _traj_case(S):
    cfg = ounet.UNetConfig(model_channels=64)
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 9)
    b, side = 4, 16
    inputs = dict(x_T=W.synth_input("smp.x_T", (b, 4, side, side)), inp=W.synth_input("smp.inpaint", (b, 4, side, side)),
                  msk=(W.synth_input("smp.mask", (b, 1, side, side)) > 0).float(),
                  cond=W.synth_input("smp.cond", (b, 2, 768)), uc=W.synth_input("smp.uc", (1, 2, 768)).repeat(b, 1, 1))
    return cfg, sd, inputs, osampler.Schedule(S)


@functools.lru_cache(maxsize=None)
def _traj_ref(S, scale):
    _threads()
    cfg, sd, i, sch = _traj_case(S)
    eps = lambda xx, tt, cc: ounet.unet_forward(sd, cfg, xx, tt, cc)
    ref, _ = osampler.ddim_sample(eps, sch, i["cond"], i["x_T"], torch.cat([i["inp"], i["msk"]], 1), scale=scale,
                                  uncond=i["uc"], log_every_t=1000)
    return ref


def _traj_engine(S, scale, use_graph=True):
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    cfg, sd, i, sch = _traj_case(S)
    net = _unet(cfg, 16)
    net.load_state_dict(sd)
    net = net.cuda()

    class Model:
        num_timesteps = 1000
        device = torch.device("cuda")
        betas = torch.from_numpy(sch.buffers["betas"]).cuda()
        alphas_cumprod = torch.from_numpy(sch.buffers["alphas_cumprod"]).cuda()
        alphas_cumprod_prev = torch.from_numpy(sch.buffers["alphas_cumprod_prev"]).cuda()

        @staticmethod
        def apply_model(xx, tt, cc):
            return net(xx, tt, context=cc)

    s = DDIMSampler(Model(), graph=use_graph)
    got, _ = s.sample(S=S, batch_size=4, shape=[4, 16, 16], conditioning=i["cond"].cuda(), verbose=False, eta=0.0,
                      x_T=i["x_T"].cuda(), unconditional_guidance_scale=scale, unconditional_conditioning=i["uc"].cuda(),
                      log_every_t=1000, test_model_kwargs={"inpaint_image": i["inp"].cuda(),
                                                           "inpaint_mask": i["msk"].cuda()})
    return got, s


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_ddim50_final_latent(dtype, scale):
    """50 sequential steps; fp16 storage without guidance is the north star's 1e-3."""
    _set(dtype)
    got, _ = _traj_engine(50, scale)
    check(rel_l2(got.cpu(), _traj_ref(50, scale)), TOL_DDIM50[(dtype, scale)], f"ddim50_cfg{scale:g}_{dtype}")


def test_ddim250_fp16_schedule_and_latent():
    """BASELINE config 5: DDIM-250, fp16.  The timestep table is the integer golden [1, 5, ..., 997]."""
    _set(torch.float16)
    got, s = _traj_engine(250, 1.0)
    assert np.array_equal(s.ddim_timesteps, np.arange(1, 1000, 4))
    check(rel_l2(got.cpu(), _traj_ref(250, 1.0)), TOL_DDIM250[torch.float16], "ddim250_cfg1_fp16")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lidar", [False, True], ids=["camera", "lidar"])
def test_full_width_vae_256(dtype, lidar):
    """ch = 128 VAE (83.7 M / 84.3 M parameters) on one 256x256 input: encode moments, decode."""
    _set(dtype)
    _threads()
    cfg = ovae.VAEConfig(in_channels=2 if lidar else 3, out_ch=2 if lidar else 3, ch=128, lidar_adapter=lidar)
    sd = W.synth_state_dict(ovae.vae_param_shapes(cfg), 23)
    x = W.synth_input(f"prod.vae.{lidar}", (1, cfg.in_channels, 256, 256), kind="uniform")
    z = W.synth_input(f"prod.vae.z.{lidar}", (1, 4, 32, 32))
    ref_m = ovae.encode_moments(sd, cfg, x)
    ref_rec = ovae.decode(sd, cfg, z)
    vae = _vae(cfg, res=256)
    vae.load_state_dict(sd)
    vae = vae.cuda()
    tag = "lidar" if lidar else "camera"
    check(rel_l2(vae.encode(x.cuda()).parameters.cpu(), ref_m), TOL_VAE[dtype], f"vae128_{tag}_encode_{dtype}")
    check(rel_l2(vae.decode(z.cuda()).cpu(), ref_rec), TOL_VAE[dtype], f"vae128_{tag}_decode_{dtype}")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lidar", [False, True], ids=["camera", "lidar"])
def test_full_width_vae_512(dtype, lidar):
    """The VAEs as `mobi_nusc_512` / `all-classes_512` configure them (resolution 512: BASELINE configs 3-5): ch = 128, one
    512 x 512 input -- mid.attn_1 runs on 4,096 tokens of 512 channels (single head), the first / last levels on 128
    channels at 512 x 512 (the lidar adapter's 1 x 5 convolutions there).  Oracle outputs from
    tests/golden/oracle_outputs.npz (tests/oracle_cases.py vae512)."""
    _set(dtype)
    from tests import oracle_cases as oc
    cfg, x, z = oc.vae512_inputs(lidar)
    ref_m, ref_rec = oc.vae512(lidar)
    sd = W.synth_state_dict(ovae.vae_param_shapes(cfg), 23)
    vae = _vae(cfg, res=512)
    vae.load_state_dict(sd)
    vae = vae.cuda()
    tag = "lidar" if lidar else "camera"
    check(rel_l2(vae.encode(x.cuda()).parameters.cpu(), ref_m), TOL_VAE[dtype], f"vae128_512_{tag}_encode_{dtype}")
    check(rel_l2(vae.decode(z.cuda()).cpu(), ref_rec), TOL_VAE[dtype], f"vae128_512_{tag}_decode_{dtype}")


@pytest.mark.parametrize("lidar", [False, True], ids=["camera", "lidar"])
def test_vae_decoder_precision_levels(lidar):
    """The fp16 decoders' operand forms (model.py precise_level): 16-bit operands on fp32 streams (0), hi | lo activations against
    [W ; W] (1), hi | lo | hi against [W ; W ; W - T(W)] (2, the default with fp16 storage) on the ch = 128 decoder at 256 x 256 against
    the oracle's decode: every level is a launch-for-launch different path, each has to beat the one before by what the split buys
    (measured: 7.9e-4 -> 5.6e-4 -> 1.8e-4 camera, 9.2e-4 -> 6.2e-4 -> 1.8e-4 range view at 512 x 512, profiles/r05_decoder_err.txt)."""
    from mobi_amd.ldm.modules.diffusionmodules import model as M
    _set(torch.float16)
    _threads()
    cfg = ovae.VAEConfig(in_channels=2 if lidar else 3, out_ch=2 if lidar else 3, ch=128, lidar_adapter=lidar)
    sd = W.synth_state_dict(ovae.vae_param_shapes(cfg), 23)
    z = W.synth_input(f"prod.vae.z.{lidar}", (1, 4, 32, 32))
    ref_rec = ovae.decode(sd, cfg, z)
    vae = _vae(cfg, res=256)
    vae.load_state_dict(sd)
    vae = vae.cuda()
    was = M._PRECISE_ENV
    errs = []
    try:
        for level in ("0", "1", "2"):
            M._PRECISE_ENV = level
            errs.append(rel_l2(vae.decode(z.cuda()).cpu(), ref_rec))
            record(f"vae128_{'lidar' if lidar else 'camera'}_decode_precise{level}", errs[-1])
    finally:
        M._PRECISE_ENV = was
    print("decoder error by operand form:", [f"{e:.3e}" for e in errs])
    assert errs[1] < 0.85 * errs[0] and errs[2] < 0.5 * errs[1], errs
    assert errs[2] < 4e-4, errs


def test_pingpong_race_screen():
    """tools/race_screen.py: seven ping-pong shapes x 60 repeats with odd persistent-block counts and an HBM-thrashing
    copy in between, every output bit for bit equal to the first."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("race_screen", os.path.join(root, "tools", "race_screen.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.screen(repeats=60) == 0


# ---- end to end at production width, pixel space (BASELINE config 1's workload; north star: 1e-3 rel-L2) ---------------
# measured on the MI355X (profiles/r05_error_table.txt, profiles/r05_parity.json): fp16 (the decoders on fp32 streams with every
# convolution's operands split, activations and weights: the defaults of that storage type, model.py precise_level) latent 4.5e-4 /
# 4.8e-4 (64 x 64 / 32 x 32), camera picture 3.7e-4 / 4.6e-4, range view 7.0e-4 / 6.9e-4 (round 4: 9.1e-4 / 9.8e-4 and 1.24 /
# 1.37e-3); bf16 latent 3.8e-3 / 4.2e-3, pictures 0.98 - 1.42e-2.  ALL THREE fp16 bounds are the north star's 1e-3 itself (the
# arithmetic is bit-reproducible: fixed-order reductions).  profiles/r05_decoder_err.txt: on the ORACLE's latent the lidar decoder is
# at 1.8e-4 (16-bit trunk 1.57e-3, fp32 trunk 1.08e-3, fp32 streams 9.2e-4, + precise tail 8.8e-4, activations split everywhere
# 6.2e-4, weights too 1.8e-4); what is left end to end is the sampler's latent error, which the lidar decoder amplifies 1.6x.
TOL_E2E = {(torch.float16, "latent"): 1e-3, (torch.float16, "pixel_camera"): 1e-3, (torch.float16, "pixel_range"): 1e-3,
           (torch.bfloat16, "latent"): 8e-3, (torch.bfloat16, "pixel_camera"): 2.8e-2, (torch.bfloat16, "pixel_range"): 2.8e-2}


def _write_parity(side, dtype, numbers, kind=None):
    """The measured end-to-end numbers of this run, tied to the library they were measured with: gpurun_out/parity_last.json
    (scratch; tools/collect_profiles.py copies it to profiles/<tag>_parity.json with the commit), which bench.py reports as its
    `parity` / `meets_north_star_tolerance` records -- so the bench line can say whether its library is the measured one."""
    import hashlib
    import json
    from mobi_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_last.json")
    with open(build.LIB, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("lib_sha16") != sha:
            doc = {}
    except (OSError, ValueError):
        doc = {}
    doc["lib_sha16"] = sha
    doc["src_sha16"] = build.sources_sha16()
    doc["source"] = ("tests/test_gpu_production.py::test_end_to_end_pixel_space on the MI355X: get_input -> DDIM-10 -> decode_sample -> "
                     "decode_first_stage + clamp at FULL width (1.04 B-parameter UNet, ch = 128 VAEs), one object, against the CPU "
                     "oracle's run of the same sequence (tests/oracle_cases.py e2e, oracle_outputs.npz); rel-L2")
    key = ("mobi_nusc-mini_256 (BASELINE config 1 workload: latent 32 x 32)" if side == 32
           else "mobi_nusc_512 (one camera / lidar pair: latent 64 x 64)")
    # kind None: the DDIM-10 case; "ddim50" / "plms50_cfg5": the 50-step cases of test_end_to_end_pixel_space_50_steps
    name = ("fp16" if dtype == torch.float16 else "bf16") + (f"_{kind}" if kind else "")
    doc.setdefault(key, {})[name] = {k: float(f"{v:.4g}") for k, v in numbers.items()}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


class _TokenStage(torch.nn.Module):
    """Stands for the conditioning stage (CLIP tower + bbox embedder, pinned on their own in tests/test_gpu_cond_producer.py):
    hands `get_learned_conditioning` the tokens the oracle case fixes."""

    def __init__(self, toks):
        super().__init__()
        self.toks, self.calls = toks, 0

    def encode(self, c):
        m = ("cam", "lidar")[self.calls % 2]                # get_input encodes the camera's conditioning, then the lidar's
        self.calls += 1
        return {"ref_image_token": self.toks[f"tok_{m}"].cuda(), "ref_bbox_token": self.toks[f"bbox_{m}"].cuda()}


@functools.lru_cache(maxsize=None)
def _e2e_vae_sd(lidar):
    from tests import oracle_cases as oc
    vcfg = ovae.VAEConfig(in_channels=2 if lidar else 3, out_ch=2 if lidar else 3, ch=128, lidar_adapter=lidar)
    return W.synth_state_dict(ovae.vae_param_shapes(vcfg), oc.E2E_SEEDS["vae"])


@functools.lru_cache(maxsize=None)
def _e2e_model(side):
    """LatentDiffusion from the shipped config of that resolution; the 1.04 B-parameter UNet is the cached full-width net of
    this file (same synthetic parameters, seed 13: generating them takes minutes), built small in the constructor and swapped."""
    from tests import oracle_cases as oc
    from mobi_amd.ldm.util import instantiate_from_config, load_config
    assert oc.E2E_SEEDS["unet"] == 13
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "mobi_nusc-mini_256.yaml" if side == 32 else "mobi_nusc_512.yaml"
    cfg = load_config(os.path.join(root, "configs", name), ["model.params.lidar_stage_config.params.ckpt_path=null"])
    cfg["model"]["params"]["cond_stage_config"] = "__is_unconditional__"
    full_mc = cfg["model"]["params"]["unet_config"]["params"]["model_channels"]
    cfg["model"]["params"]["unet_config"]["params"]["model_channels"] = 32
    model = instantiate_from_config(cfg["model"])
    assert full_mc == ounet.UNetConfig().model_channels and model.image_size == side
    model.model.diffusion_model = _full_width_net()
    model.first_stage_model.load_state_dict(_e2e_vae_sd(False))
    model.lidar_stage_model.load_state_dict(_e2e_vae_sd(True))
    i = oc.e2e_inputs(side)
    with torch.no_grad():
        model.proj_out.weight.copy_(i["proj_w"])
        model.proj_out.bias.copy_(i["proj_b"])
    return model.cuda().eval(), i


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("side", [32, 64], ids=["mini256", "nusc512_pair"])
def test_end_to_end_pixel_space(dtype, side):
    """scripts/inference_test_bench.py:416-464 on the engine at FULL width for one object -- get_input (four VAE encodes,
    conditioning projection, lidar alignment, interleave) -> DDIMSampler.sample(S = 10) -> decode_sample -> decode_first_stage
    + clamp -- against the CPU oracle's run of the same sequence (tests/oracle_cases.py e2e; oracle_outputs.npz): the 9-channel
    input's index parts bit-exact, the final latent and the decoded camera picture / range view in rel-L2."""
    from tests import oracle_cases as oc
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    _set(dtype)
    model, i = _e2e_model(side)
    ref = oc.e2e(side)
    model.cond_stage_model = _TokenStage(i)
    batch = {"image": {"GT": i["img"], "inpaint_image": i["img"] * i["mask"], "inpaint_mask": i["mask"],
                       "cond": {"ref_image": torch.zeros(1, 3, 224, 224), "ref_bbox": torch.zeros(1, 8, 3)}},
             "lidar": {"range_data": i["rng"], "range_data_inpaint": i["rng"] * i["mask"], "range_mask": i["mask"],
                       "cond": {"ref_image": torch.zeros(1, 3, 224, 224), "ref_bbox": torch.zeros(1, 8, 3)}}}
    to_dev = lambda d: {k: to_dev(v) if isinstance(v, dict) else v.cuda() for k, v in d.items()}
    noises = {"cam_gt": i["n_cam_gt"].cuda(), "cam_inpaint": i["n_cam_inpaint"].cuda(), "lidar_gt": i["n_lidar_gt"].cuda(),
              "lidar_inpaint": i["n_lidar_inpaint"].cuda()}
    data = model.get_input(to_dev(batch), model.first_stage_key, force_c_encode=True, noises=noises)
    z, cond = data["z"], data["cond"]
    assert z.shape == ref["z"].shape and cond.shape == ref["cond"].shape
    assert torch.equal(z[:, 8].cpu(), ref["z"][:, 8])                                   # nearest-resized mask: index-only
    tag = f"e2e_{side}_{dtype}"
    check(rel_l2(z[:, :8].cpu(), ref["z"][:, :8]), TOL_VAE[dtype], tag + "_encode")
    check(rel_l2(cond.cpu(), ref["cond"]), {torch.float16: 5e-4, torch.bfloat16: 2.4e-3}[dtype], tag + "_cond")   # 1.4e-4 / 1.2e-3
    samples, _ = DDIMSampler(model).sample(S=oc.E2E_STEPS, batch_size=2, shape=[4, side, side], conditioning=cond, verbose=False,
                                           eta=0.0, x_T=i["x_T"].cuda(),
                                           test_model_kwargs={"inpaint_image": z[:, 4:8].contiguous(),
                                                              "inpaint_mask": z[:, 8:9].contiguous()})
    lat = rel_l2(samples.cpu(), ref["samples"])
    h_cam, h_lid = model.decode_sample(samples, data["z_lidar"])
    image = model.decode_first_stage(h_cam.contiguous(), clamp=(-1., 1.))
    rng = model.decode_first_stage(h_lid.contiguous(), module_name="lidar_stage_model", clamp=(-1., 1.))
    assert image.shape == ref["image"].shape and rng.shape == ref["range"].shape
    pix_c, pix_r = rel_l2(image.float().cpu(), ref["image"]), rel_l2(rng.float().cpu(), ref["range"])
    _write_parity(side, dtype, {"encode_rel_l2": rel_l2(z[:, :8].cpu(), ref["z"][:, :8]), "cond_rel_l2": rel_l2(cond.cpu(), ref["cond"]),
                                "latent_rel_l2": lat, "pixel_rel_l2_camera": pix_c, "pixel_rel_l2_range": pix_r})
    check(lat, TOL_E2E[(dtype, "latent")], tag + "_latent")
    check(pix_c, TOL_E2E[(dtype, "pixel_camera")], tag + "_pixel_camera")
    check(pix_r, TOL_E2E[(dtype, "pixel_range")], tag + "_pixel_range")


# ---- the same at the shipped invocation's length: 50 steps (VERDICT r04 "missing" #4) ---------------------------------------
# (dtype, kind, quantity) -> bound: at most 2x the value measured on the MI355X (profiles/r05_error_table.txt); ALL fp16 DDIM-50
# bounds are the north star's 1e-3 itself.  Measured, fp16: DDIM-50 latent 4.3e-4 / 3.9e-4 (32 x 32 / 64 x 64), camera picture
# 4.4e-4 / 3.5e-4, range view 6.6e-4 / 6.2e-4; PLMS-50 at guidance 5 latent 5.8e-4, camera 4.6e-4, range view 1.15e-3 (guidance
# multiplies the difference of two UNet evaluations by 5: the one case over 1e-3, by the latent's error alone).  bf16: 3.4 - 5.0e-3 /
# 0.9 - 2.0e-2.  (The reference pictures of these cases are stored in fp16: ~1.6e-4 of rounding noise that counts against the engine.)
TOL_E2E_LONG = {
    (torch.float16, "ddim50"): dict(latent=1.0e-3, pixel_camera=1.0e-3, pixel_range=1.0e-3),
    (torch.bfloat16, "ddim50"): dict(latent=7.3e-3, pixel_camera=2.1e-2, pixel_range=2.75e-2),
    (torch.float16, "plms50_cfg5"): dict(latent=1.2e-3, pixel_camera=1.0e-3, pixel_range=2.3e-3),
    (torch.bfloat16, "plms50_cfg5"): dict(latent=1.0e-2, pixel_camera=1.9e-2, pixel_range=4.0e-2),
}


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("side,kind", [(32, "ddim50"), (32, "plms50_cfg5"), (64, "ddim50")],
                         ids=["mini256_ddim50", "mini256_plms50_cfg5", "nusc512_pair_ddim50"])
def test_end_to_end_pixel_space_50_steps(dtype, side, kind):
    """`test_end_to_end_pixel_space` at the length the shipped scripts sample with (scripts/realism_test_bench.sh:95-102:
    `--ddim_steps 50`): DDIM-50 at guidance 1 (the harness default) at both resolutions, and PLMS-50 at guidance 5 (`--plms
    --scale 5`, what the scripts run) on the 256 case, full width, one object, against the CPU oracle (oracle_cases.e2e_long)."""
    from tests import oracle_cases as oc
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    from mobi_amd.ldm.models.diffusion.plms import PLMSSampler
    _set(dtype)
    model, i = _e2e_model(side)
    ref = oc.e2e_long(side, kind)
    spec = oc.E2E_LONG[kind]
    model.cond_stage_model = _TokenStage(i)
    batch = {"image": {"GT": i["img"], "inpaint_image": i["img"] * i["mask"], "inpaint_mask": i["mask"],
                       "cond": {"ref_image": torch.zeros(1, 3, 224, 224), "ref_bbox": torch.zeros(1, 8, 3)}},
             "lidar": {"range_data": i["rng"], "range_data_inpaint": i["rng"] * i["mask"], "range_mask": i["mask"],
                       "cond": {"ref_image": torch.zeros(1, 3, 224, 224), "ref_bbox": torch.zeros(1, 8, 3)}}}
    to_dev = lambda d: {k: to_dev(v) if isinstance(v, dict) else v.cuda() for k, v in d.items()}
    noises = {"cam_gt": i["n_cam_gt"].cuda(), "cam_inpaint": i["n_cam_inpaint"].cuda(), "lidar_gt": i["n_lidar_gt"].cuda(),
              "lidar_inpaint": i["n_lidar_inpaint"].cuda()}
    data = model.get_input(to_dev(batch), model.first_stage_key, force_c_encode=True, noises=noises)
    z, cond = data["z"], data["cond"]
    uc = oc.e2e_uncond(side).cuda() if spec["scale"] != 1.0 else None
    if spec["sampler"] == "ddim":
        samples, _ = DDIMSampler(model).sample(S=spec["steps"], batch_size=2, shape=[4, side, side], conditioning=cond, verbose=False,
                                               eta=0.0, x_T=i["x_T"].cuda(), unconditional_guidance_scale=spec["scale"],
                                               unconditional_conditioning=uc,
                                               test_model_kwargs={"inpaint_image": z[:, 4:8].contiguous(),
                                                                  "inpaint_mask": z[:, 8:9].contiguous()})
    else:
        samples, _ = PLMSSampler(model).sample(S=spec["steps"], batch_size=2, shape=[4, side, side], conditioning=cond, verbose=False,
                                               x_T=i["x_T"].cuda(), unconditional_guidance_scale=spec["scale"],
                                               unconditional_conditioning=uc, inpaint_image=z[:, 4:8].contiguous(),
                                               inpaint_mask=z[:, 8:9].contiguous())
    lat = rel_l2(samples.cpu(), ref["samples"])
    h_cam, h_lid = model.decode_sample(samples, data["z_lidar"])
    image = model.decode_first_stage(h_cam.contiguous(), clamp=(-1., 1.))
    rng = model.decode_first_stage(h_lid.contiguous(), module_name="lidar_stage_model", clamp=(-1., 1.))
    pix_c, pix_r = rel_l2(image.float().cpu(), ref["image"]), rel_l2(rng.float().cpu(), ref["range"])
    tol = TOL_E2E_LONG[(dtype, kind)]
    tag = f"e2e_{side}_{kind}_{dtype}"
    _write_parity(side, dtype, {"latent_rel_l2": lat, "pixel_rel_l2_camera": pix_c, "pixel_rel_l2_range": pix_r}, kind=kind)
    check(lat, tol["latent"], tag + "_latent")
    check(pix_c, tol["pixel_camera"], tag + "_pixel_camera")
    check(pix_r, tol["pixel_range"], tag + "_pixel_range")
