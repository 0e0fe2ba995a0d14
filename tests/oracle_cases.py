"""CPU-oracle outputs of the expensive GPU parity cases, computed ONCE: `tests/golden/oracle_outputs.npz` holds what these
functions return (written by `python tests/golden/make_oracle_outputs.py`, minutes of CPU time), the `-m gpu` tests read
it instead of spending ~5 of their 10 minutes in the oracle on the GPU box's host cores, and
`tests/test_oracle_outputs_cpu.py` recomputes a sample of it live so that the file cannot drift from the oracle.
A key that is missing from the file is computed live.  Test infrastructure only (imports `oracle/`)."""
import functools
import os

import numpy as np
import torch

from oracle import sampler as osampler, unet as ounet, vae as ovae, weights as W

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_outputs.npz")
LIVE = os.environ.get("MOBI_ORACLE_LIVE") == "1"          # ignore the file (the generator, the CPU check)


@functools.lru_cache(maxsize=None)
def _file():
    return dict(np.load(PATH)) if os.path.exists(PATH) and not LIVE else {}


def _threads():
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))


PROD_T = [981, 981, 741, 741, 501, 501, 261, 261, 21, 21, 1, 1, 901, 901, 481, 481]


def prod_inputs(side, n):
    return (W.synth_input(f"prod.x{side}", (n, 9, side, side)), W.synth_input(f"prod.c{side}", (n, 2, 768)),
            torch.tensor(PROD_T[:n], dtype=torch.long))


def prod_forward(side, n, pairs=None, live=False):
    """Full-width UNet (seed 13) on the production batch, pair by pair -> [n, 4, side, side] (or the given pairs only)."""
    key = f"prod_{side}_{n}"
    if not live and pairs is None and key in _file():
        return torch.from_numpy(_file()[key])
    _threads()
    cfg = ounet.UNetConfig()
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 13)
    x, ctx, t = prod_inputs(side, n)
    idx = range(0, n, 2) if pairs is None else [2 * p for p in pairs]
    return torch.cat([ounet.unet_forward(sd, cfg, x[i:i + 2], t[i:i + 2], ctx[i:i + 2]) for i in idx])


def full_width16(live=False):
    """Full-width UNet (seed 5), one pair at 16 x 16, t = 741."""
    if not live and "full_width16" in _file():
        return torch.from_numpy(_file()["full_width16"])
    _threads()
    cfg = ounet.UNetConfig()
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 5)
    return ounet.unet_forward(sd, cfg, W.synth_input("uf.x", (2, 9, 16, 16)), torch.tensor([741, 741], dtype=torch.long),
                              W.synth_input("uf.c", (2, 2, 768)))


def traj_inputs():
    b, side = 4, 16
    return dict(x_T=W.synth_input("smp.x_T", (b, 4, side, side)), inp=W.synth_input("smp.inpaint", (b, 4, side, side)),
                msk=(W.synth_input("smp.mask", (b, 1, side, side)) > 0).float(), cond=W.synth_input("smp.cond", (b, 2, 768)),
                uc=W.synth_input("smp.uc", (1, 2, 768)).repeat(b, 1, 1), x0=W.synth_input("smp.x0", (b, 4, side, side)),
                cmask=(W.synth_input("smp.cmask", (b, 1, side, side)) > 0).float(),
                mn=W.synth_input("smp.mn", (10, b, 4, side, side)), sn=W.synth_input("smp.sn", (10, b, 4, side, side)))


def trajectories10(live=False, only=None):
    """DDIM-10 / PLMS-10 finals with and without guidance, the DDIM intermediates' count, and mask-mode DDIM (eta 1) on the
    reduced UNet (model_channels 64, seed 9): dict of tensors."""
    keys = ["ddim_1.0", "ddim_5.0", "plms_1.0", "plms_5.0", "mask_eta1", "n_pred_x0"]
    if not live and all("traj10_" + k in _file() for k in keys):
        return {k: torch.from_numpy(_file()["traj10_" + k]) for k in keys}
    _threads()
    cfg = ounet.UNetConfig(model_channels=64)
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 9)
    i = traj_inputs()
    eps = lambda x, t, c: ounet.unet_forward(sd, cfg, x, t, c)
    rest = torch.cat([i["inp"], i["msk"]], 1)
    sch = osampler.Schedule(10)
    out = {}
    for scale in (1.0, 5.0):
        if only and f"ddim_{scale}" not in only:
            continue
        ref, rint = osampler.ddim_sample(eps, sch, i["cond"], i["x_T"], rest, scale=scale, uncond=i["uc"], log_every_t=3)
        out[f"ddim_{scale}"] = ref
        out["n_pred_x0"] = torch.tensor(len(rint["pred_x0"]))
        if only:
            continue
        out[f"plms_{scale}"], _ = osampler.plms_sample(eps, sch, i["cond"], i["x_T"], rest, scale=scale, uncond=i["uc"], log_every_t=3)
    if not only:
        out["mask_eta1"], _ = osampler.ddim_sample(eps, osampler.Schedule(10, eta=1.0), i["cond"], i["x_T"], rest, mask=i["cmask"],
                                                   x0=i["x0"], mask_noise=i["mn"], step_noise=i["sn"])
    return out


def vae512_inputs(lidar):
    cfg = ovae.VAEConfig(in_channels=2 if lidar else 3, out_ch=2 if lidar else 3, ch=128, lidar_adapter=lidar)
    x = W.synth_input(f"prod.vae512.{lidar}", (1, cfg.in_channels, 512, 512), kind="uniform")
    z = W.synth_input(f"prod.vae512.z.{lidar}", (1, 4, 64, 64))
    return cfg, x, z


def vae512(lidar, live=False):
    """ch = 128 VAE (seed 23) at the resolution `mobi_nusc_512` / `all-classes_512` run it (512 x 512: mid.attn_1 sees 4,096
    tokens of 512 channels): encoder moments [1, 8, 64, 64] fp32 and the decoded picture (kept as fp16 in the file: 2.4e-4
    of quantisation against tolerances of 5e-3 and more) -> (moments, picture)."""
    tag = "lidar" if lidar else "camera"
    km, kd = f"vae512_{tag}_moments", f"vae512_{tag}_decode"
    if not live and km in _file() and kd in _file():
        return torch.from_numpy(_file()[km]), torch.from_numpy(_file()[kd].astype(np.float32))
    _threads()
    cfg, x, z = vae512_inputs(lidar)
    sd = W.synth_state_dict(ovae.vae_param_shapes(cfg), 23)
    return ovae.encode_moments(sd, cfg, x), ovae.decode(sd, cfg, z)


# ---- end to end at production width (BASELINE config 1's workload; VERDICT r03 "missing" #3) ------------------------
E2E_SEEDS = dict(unet=13, vae=23)
E2E_STEPS = 10


def e2e_inputs(side):
    """Synthetic A0 batch for ONE object (camera + lidar) at R = 8 * side, the explicit noises, and the tokens a
    conditioning stage would hand to `get_learned_conditioning` (the CLIP tower / bbox embedder are pinned on their own:
    tests/test_gpu_cond_producer.py)."""
    R = 8 * side
    mask = torch.ones(1, 1, R, R)
    mask[:, :, R // 4: 3 * R // 4, R // 4: 3 * R // 4] = 0            # centred 50 %-side hole
    d = {"R": R, "mask": mask,
         "img": W.synth_input(f"e2e{side}.img", (1, 3, R, R), kind="uniform"),
         "rng": W.synth_input(f"e2e{side}.rng", (1, 2, R, R), kind="uniform"),
         "x_T": W.synth_input(f"e2e{side}.x_T", (2, 4, side, side)),
         "proj_w": torch.from_numpy(W.synth_param("proj_out.weight", (768, 1024), 31)),
         "proj_b": torch.from_numpy(W.synth_param("proj_out.bias", (768,), 31))}
    for m in ("cam", "lidar"):
        d[f"tok_{m}"] = W.synth_input(f"e2e{side}.tok.{m}", (1, 1, 1024))
        d[f"bbox_{m}"] = W.synth_input(f"e2e{side}.bboxtok.{m}", (1, 1, 768))
        for k in ("gt", "inpaint"):
            d[f"n_{m}_{k}"] = W.synth_input(f"e2e{side}.n.{m}.{k}", (1, 4, side, side))
    return d


def e2e(side, live=False):
    """`get_input -> DDIMSampler.sample(S=10, eta 0, scale 1) -> decode_sample -> decode_first_stage + clamp` of
    scripts/inference_test_bench.py:416-464 on the CPU oracle at FULL width (1.04 B-parameter UNet, ch = 128 VAEs), one
    object: side 32 = `mobi_nusc-mini_256` (BASELINE config 1), side 64 = one pair of `mobi_nusc_512`.
    -> dict(z [2,9,s,s], cond [2,2,768], samples [2,4,s,s], image [1,3,R,R], range [1,2,R,R])."""
    keys = ["z", "cond", "samples", "image", "range"]
    if not live and all(f"e2e{side}_{k}" in _file() for k in keys):
        return {k: torch.from_numpy(_file()[f"e2e{side}_{k}"].astype(np.float32)) for k in keys}
    from oracle import pipeline
    import torch.nn.functional as F
    _threads()
    i = e2e_inputs(side)
    ucfg = ounet.UNetConfig()
    usd = W.synth_state_dict(ounet.unet_param_shapes(ucfg), E2E_SEEDS["unet"])
    cam_cfg = ovae.VAEConfig(in_channels=3, out_ch=3, ch=128, lidar_adapter=False)
    lid_cfg = ovae.VAEConfig(in_channels=2, out_ch=2, ch=128, lidar_adapter=True)
    cam_sd = W.synth_state_dict(ovae.vae_param_shapes(cam_cfg), E2E_SEEDS["vae"])
    lid_sd = W.synth_state_dict(ovae.vae_param_shapes(lid_cfg), E2E_SEEDS["vae"])
    with torch.no_grad():
        z_image = pipeline.encode_modality(cam_sd, cam_cfg, i["img"], i["img"] * i["mask"], i["mask"], i["n_cam_gt"],
                                           i["n_cam_inpaint"], 0.18215)
        z_lidar = pipeline.encode_modality(lid_sd, lid_cfg, i["rng"], i["rng"] * i["mask"], i["mask"], i["n_lidar_gt"],
                                           i["n_lidar_inpaint"], 0.18215)
        z_lidar_al, _ = pipeline.align_lidar(z_lidar, torch.zeros(1, 8, 3), side)
        z = pipeline.cat_interleave([z_image, z_lidar_al])
        cond = pipeline.cat_interleave([torch.cat([F.linear(i[f"tok_{m}"], i["proj_w"], i["proj_b"]), i[f"bbox_{m}"]], 1)
                                        for m in ("cam", "lidar")])
        eps = lambda x, t, c: ounet.unet_forward(usd, ucfg, x, t, c)
        samples, _ = osampler.ddim_sample(eps, osampler.Schedule(E2E_STEPS), cond, i["x_T"], z[:, 4:9].contiguous())
        h_cam, h_lid = pipeline.decode_sample(samples, z_lidar[:, :4], side)
        image = pipeline.decode_first_stage(cam_sd, cam_cfg, h_cam, 0.18215)
        rng = pipeline.decode_first_stage(lid_sd, lid_cfg, h_lid, 0.18215)
    return {"z": z, "cond": cond, "samples": samples, "image": image, "range": rng}


# ---- the same sequence at the SHIPPED invocation's length (scripts/realism_test_bench.sh:95-102: 50 steps; DDIM at guidance 1, the
# harness default, and PLMS at guidance 5, what the shipped scripts run) -- VERDICT r04 "missing" #4 -------------------------------
E2E_LONG = {"ddim50": dict(sampler="ddim", steps=50, scale=1.0), "plms50_cfg5": dict(sampler="plms", steps=50, scale=5.0)}


def e2e_uncond(side):
    """What the harness hands the samplers as `unconditional_conditioning` (inference_test_bench.py:425-431: the learnt
    `learnable_vector` and `bbox_uncond_vector`, repeated over the batch): two synthetic [1, 1, 768] tokens."""
    uc = torch.cat([W.synth_input(f"e2e{side}.uc.ref", (1, 1, 768)), W.synth_input(f"e2e{side}.uc.bbox", (1, 1, 768))], dim=1)
    return uc.repeat(2, 1, 1)


def e2e_long(side, kind, live=False):
    """`e2e(side)`'s 9-channel input, conditioning and x_T through 50 sampler steps (`kind` in E2E_LONG), then
    decode_sample -> decode_first_stage + clamp, on the CPU oracle at FULL width -> dict(samples, image, range)."""
    keys = ["samples", "image", "range"]
    if not live and all(f"e2e{side}_{kind}_{k}" in _file() for k in keys):
        return {k: torch.from_numpy(_file()[f"e2e{side}_{kind}_{k}"].astype(np.float32)) for k in keys}
    from oracle import pipeline
    _threads()
    spec = E2E_LONG[kind]
    i = e2e_inputs(side)
    base = e2e(side)                                          # z, cond of the stored short case (or computed live)
    z, cond = base["z"], base["cond"]
    ucfg = ounet.UNetConfig()
    usd = W.synth_state_dict(ounet.unet_param_shapes(ucfg), E2E_SEEDS["unet"])
    cam_cfg = ovae.VAEConfig(in_channels=3, out_ch=3, ch=128, lidar_adapter=False)
    lid_cfg = ovae.VAEConfig(in_channels=2, out_ch=2, ch=128, lidar_adapter=True)
    cam_sd = W.synth_state_dict(ovae.vae_param_shapes(cam_cfg), E2E_SEEDS["vae"])
    lid_sd = W.synth_state_dict(ovae.vae_param_shapes(lid_cfg), E2E_SEEDS["vae"])
    with torch.no_grad():
        # the un-cropped lidar latent decode_sample pastes the sample back into: the lidar GT encode of e2e()
        z_lidar = pipeline.encode_modality(lid_sd, lid_cfg, i["rng"], i["rng"] * i["mask"], i["mask"], i["n_lidar_gt"],
                                           i["n_lidar_inpaint"], 0.18215)
        eps = lambda x, t, c: ounet.unet_forward(usd, ucfg, x, t, c)
        fn = osampler.ddim_sample if spec["sampler"] == "ddim" else osampler.plms_sample
        uc = e2e_uncond(side) if spec["scale"] != 1.0 else None
        samples, _ = fn(eps, osampler.Schedule(spec["steps"]), cond, i["x_T"], z[:, 4:9].contiguous(), scale=spec["scale"], uncond=uc)
        h_cam, h_lid = pipeline.decode_sample(samples, z_lidar[:, :4], side)
        image = pipeline.decode_first_stage(cam_sd, cam_cfg, h_cam, 0.18215)
        rng = pipeline.decode_first_stage(lid_sd, lid_cfg, h_lid, 0.18215)
    return {"samples": samples, "image": image, "range": rng}
