"""CPU: the oracle (oracle/*.py) against every golden vector produced by the
reference's own modules (tests/golden/make_golden.py).  This is what pins the
oracle; the GPU tests then compare the HIP engine with the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline, sampler, schedule, unet as ounet, vae as ovae, weights as W
from tests.golden_cases import (OPS_SEED, UNET_CFGS, UNET_SEED, VAE_CFGS, VAE_SEED, load, op_sd, rel_l2)

TOL = 2e-5      # fp32 CPU vs fp32 CPU; different op order only in stack/reshape


def test_schedule_tables_bit_exact():
    g = load("schedule_tables")
    assert np.array_equal(schedule.linear_betas(), g["betas_f64"].numpy())
    for S in (10, 50, 250, 30):
        ts = schedule.ddim_timesteps(S)
        assert ts.dtype == np.int64 and np.array_equal(ts, g[f"ddim_timesteps_S{S}"].numpy())
    assert len(schedule.ddim_timesteps(30)) == 31          # the 1000 // S quirk (util.py:48-49)
    buf = schedule.ddpm_buffers()
    for k, v in buf.items():
        assert np.array_equal(v, g["ddpm_" + k].numpy()), k
    for S in (10, 50):
        for eta in (0.0, 1.0):
            p = schedule.ddim_parameters(buf["alphas_cumprod"], schedule.ddim_timesteps(S), eta)
            tag = f"S{S}_eta{int(eta)}"
            for mine, ref in (("alphas", "ddim_alphas"), ("alphas_prev", "ddim_alphas_prev"),
                              ("sigmas", "ddim_sigmas"), ("sqrt_one_minus_alphas", "ddim_sqrt_one_minus_alphas")):
                r = g[f"{ref}_{tag}"].numpy()
                assert p[mine].dtype == r.dtype and np.array_equal(p[mine], r), (mine, tag)


def test_timestep_embedding():
    g = load("ops")
    assert torch.equal(ounet.timestep_embedding(g["temb_t"], 320), g["temb_320"])
    assert torch.equal(ounet.timestep_embedding(g["temb_t"], 32), g["temb_32"])


def _shapes(prefix, all_shapes):
    return {k[len(prefix):]: s for k, s in all_shapes.items() if k.startswith(prefix)}


def test_unet_operators():
    g = load("ops")
    emb = W.synth_input("ops.emb", (4, 128))
    for tag, cin, cout in (("res64", 64, 64), ("res96to64", 96, 64)):
        shapes = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,),
                  "in_layers.2.weight": (cout, cin, 3, 3), "in_layers.2.bias": (cout,),
                  "emb_layers.1.weight": (cout, 128), "emb_layers.1.bias": (cout,),
                  "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,),
                  "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,)}
        if cin != cout:
            shapes.update({"skip_connection.weight": (cout, cin, 1, 1), "skip_connection.bias": (cout,)})
        sd = {"r." + k: v for k, v in op_sd(tag + ".", shapes).items()}
        y = ounet.res_block(sd, "r", W.synth_input(f"ops.{tag}.x", (4, cin, 8, 8)), emb)
        assert rel_l2(y, g[tag + "_y"]) < TOL
    sd = op_sd("down64.", {"op.weight": (64, 64, 3, 3), "op.bias": (64,)})
    y = F.conv2d(W.synth_input("ops.down64.x", (2, 64, 8, 8)), sd["op.weight"], sd["op.bias"], stride=2, padding=1)
    assert rel_l2(y, g["down64_y"]) < TOL
    sd = op_sd("up64.", {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)})
    y = F.conv2d(F.interpolate(W.synth_input("ops.up64.x", (2, 64, 4, 4)), scale_factor=2, mode="nearest"),
                 sd["conv.weight"], sd["conv.bias"], padding=1)
    assert rel_l2(y, g["up64_y"]) < TOL

    # SpatialTransformer with both adapters on: reuse the UNet shape table for one ST of width 64
    cfg = ounet.UNetConfig(model_channels=64, channel_mult=(1,), attention_resolutions=(1,), num_res_blocks=1)
    st_shapes = _shapes("input_blocks.1.1.", ounet.unet_param_shapes(cfg))
    sd = {"st." + k: v for k, v in op_sd("st64.", st_shapes).items()}
    y = ounet.spatial_transformer(sd, "st", W.synth_input("ops.st64.x", (4, 64, 8, 8)),
                                  W.synth_input("ops.st64.ctx", (4, 2, 768)), cfg)
    assert rel_l2(y, g["st64_y"]) < TOL

    for tk in (1, 2, 16):
        shapes = {"to_q.weight": (64, 64), "to_k.weight": (64, 48), "to_v.weight": (64, 48),
                  "to_out.0.weight": (64, 64), "to_out.0.bias": (64,)}
        sd = {"a." + k: v for k, v in op_sd(f"xattn{tk}.", shapes).items()}
        y = ounet.cross_attention(sd, "a", W.synth_input(f"ops.xattn{tk}.x", (2, 16, 64)),
                                  W.synth_input(f"ops.xattn{tk}.ctx", (2, tk, 48)), 8)
        assert rel_l2(y, g[f"xattn{tk}_y"]) < TOL
    sd = op_sd("ff64.", {"net.0.proj.weight": (512, 64), "net.0.proj.bias": (512,),
                         "net.2.weight": (64, 256), "net.2.bias": (64,)})
    x = W.synth_input("ops.ff64.x", (2, 16, 64))
    a, gate = F.linear(x, sd["net.0.proj.weight"], sd["net.0.proj.bias"]).chunk(2, dim=-1)
    y = F.linear(a * F.gelu(gate), sd["net.2.weight"], sd["net.2.bias"])
    assert rel_l2(y, g["ff64_y"]) < TOL


def test_vae_operators():
    g = load("ops")
    for tag, kh, kw in (("vres3", 3, 3), ("vres15", 1, 5)):
        shapes = {"norm1.weight": (32,), "norm1.bias": (32,), "conv1.weight": (64, 32, kh, kw), "conv1.bias": (64,),
                  "norm2.weight": (64,), "norm2.bias": (64,), "conv2.weight": (64, 64, kh, kw), "conv2.bias": (64,),
                  "nin_shortcut.weight": (64, 32, 1, 1), "nin_shortcut.bias": (64,)}
        sd = {"r." + k: v for k, v in op_sd(tag + ".", shapes).items()}
        y = ovae.resnet_block(sd, "r", W.synth_input(f"ops.{tag}.x", (2, 32, 8, 8)))
        assert rel_l2(y, g[tag + "_y"]) < TOL
    shapes = {"norm.weight": (64,), "norm.bias": (64,)}
    for n in ("q", "k", "v", "proj_out"):
        shapes.update({f"{n}.weight": (64, 64, 1, 1), f"{n}.bias": (64,)})
    sd = {"a." + k: v for k, v in op_sd("vattn.", shapes).items()}
    assert rel_l2(ovae.attn_block(sd, "a", W.synth_input("ops.vattn.x", (2, 64, 8, 8))), g["vattn_y"]) < TOL
    sd = op_sd("vdown.", {"conv.weight": (32, 32, 3, 3), "conv.bias": (32,)})
    x = F.pad(W.synth_input("ops.vdown.x", (2, 32, 8, 8)), (0, 1, 0, 1))
    assert rel_l2(F.conv2d(x, sd["conv.weight"], sd["conv.bias"], stride=2), g["vdown_y"]) < TOL
    mom = W.synth_input("ops.dg.moments", (2, 8, 4, 4)) * 3.0
    noise = W.synth_input("ops.dg.noise", (2, 4, 4, 4))
    assert torch.equal(ovae.posterior_sample(mom, noise), g["dg_sample"])
    assert torch.equal(torch.chunk(mom, 2, dim=1)[0], g["dg_mode"])


@pytest.mark.parametrize("name", list(UNET_CFGS))
def test_unet_forward(name):
    cfg, batch, side = UNET_CFGS[name]
    g = load(name)
    shapes = ounet.unet_param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    sd = W.synth_state_dict(shapes, UNET_SEED)
    y = ounet.unet_forward(sd, cfg, g["x"], g["t"], g["ctx"])
    assert y.shape == g["y"].shape
    assert rel_l2(y, g["y"]) < TOL


def test_full_size_param_count():
    """1 039 929 604 parameters for the mobi_nusc_512 UNet (probe)."""
    shapes = ounet.unet_param_shapes(ounet.UNetConfig())
    assert sum(int(np.prod(s)) for s in shapes.values()) == 1_039_929_604


@pytest.mark.parametrize("name", list(VAE_CFGS))
def test_vae(name):
    cfg = VAE_CFGS[name]
    g = load(name)
    sd = W.synth_state_dict(ovae.vae_param_shapes(cfg), VAE_SEED)
    mom = ovae.encode_moments(sd, cfg, g["x"])
    assert rel_l2(mom, g["moments"]) < TOL
    z = ovae.posterior_sample(g["moments"], g["noise"])
    assert torch.equal(z, g["z"])
    assert rel_l2(ovae.decode(sd, cfg, g["z"]), g["rec"]) < TOL


def _eps_fn():
    cfg, _, _ = UNET_CFGS["unet_mc32_mm"]
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), UNET_SEED)
    return lambda x, t, c: ounet.unet_forward(sd, cfg, x, t, c)


def test_sampler_trajectories():
    g = load("sampler")
    eps = _eps_fn()
    rest = torch.cat([g["inpaint"], g["mask"]], dim=1)
    sch = sampler.Schedule(10, eta=0.0)
    for scale in (1.0, 5.0):
        tag = f"cfg{int(scale)}"
        s, inter = sampler.ddim_sample(eps, sch, g["cond"], g["x_T"], rest, scale=scale, uncond=g["uc"], log_every_t=3)
        assert inter["ts"] == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
        assert rel_l2(s, g[f"ddim_{tag}_samples"]) < 1e-4
        assert rel_l2(torch.stack(inter["pred_x0"][1:]), g[f"ddim_{tag}_pred_x0"]) < 1e-4
        assert rel_l2(torch.stack(inter["x_inter"][1:]), g[f"ddim_{tag}_x_inter"]) < 1e-4
        s, inter = sampler.plms_sample(eps, sch, g["cond"], g["x_T"], rest, scale=scale, uncond=g["uc"], log_every_t=3)
        assert rel_l2(s, g[f"plms_{tag}_samples"]) < 1e-4
        assert rel_l2(torch.stack(inter["pred_x0"][1:]), g[f"plms_{tag}_pred_x0"]) < 1e-4


def test_mask_compositing():
    """ddim.py:145-148 with a binary mask: where mask == 1 the composited state is
    exactly q_sample(x0), where mask == 0 exactly the running sample."""
    g = load("sampler")
    eps = _eps_fn()
    rest = torch.cat([g["inpaint"], g["mask"]], dim=1)
    sch = sampler.Schedule(10, eta=1.0)
    s, inter = sampler.ddim_sample(eps, sch, g["cond"], g["x_T"], rest, mask=g["cmask"], x0=g["x0"],
                                   mask_noise=g["mask_noise"], step_noise=g["step_noise"], log_every_t=3)
    assert rel_l2(s, g["ddim_mask_samples"]) < 1e-4
    assert rel_l2(torch.stack(inter["x_inter"][1:]), g["ddim_mask_x_inter"]) < 1e-4
    # bit-exact compositing property
    ts = torch.full((4,), 901, dtype=torch.long)
    q = sampler.q_sample(sch.buffers, g["x0"], ts, g["mask_noise"][0])
    comp = q * g["cmask"] + (1.0 - g["cmask"]) * g["x_T"]
    m = g["cmask"].bool().expand_as(comp)
    assert torch.equal(comp[m], q[m]) and torch.equal(comp[~m], g["x_T"][~m])


def test_plumbing():
    g = load("plumbing")
    cam_cfg, lid_cfg = VAE_CFGS["vae_cam32"], VAE_CFGS["vae_lidar32"]
    cam = W.synth_state_dict(ovae.vae_param_shapes(cam_cfg), VAE_SEED)
    lid = W.synth_state_dict(ovae.vae_param_shapes(lid_cfg), VAE_SEED)
    z_image = pipeline.encode_modality(cam, cam_cfg, g["img"], g["img"] * g["imask"], g["imask"],
                                       g["n_cam_gt"], g["n_cam_inp"], 0.18215)
    z_lidar = pipeline.encode_modality(lid, lid_cfg, g["rng"], g["rng"] * g["rmask"], g["rmask"],
                                       g["n_lid_gt"], g["n_lid_inp"], 0.18215)
    assert rel_l2(z_image, g["z_image"]) < TOL and rel_l2(z_lidar, g["z_lidar"]) < TOL
    assert torch.equal(z_image[:, 8], g["z_image"][:, 8])          # nearest-resized mask: index-only
    h_cam, h_lid = pipeline.decode_sample(g["sample"], g["z_lidar"][:, :4], 8)
    assert torch.equal(h_cam, g["h_cam"]) and torch.equal(h_lid, g["h_lid"])
    a, b = torch.arange(6.).reshape(3, 2), -torch.arange(6.).reshape(3, 2)
    assert torch.equal(pipeline.cat_interleave([a, b]), g["cat_interleave"])
    # lidar alignment formula (ddpm.py:798-815) on a wider map: crop is centred, pad symmetric
    z = torch.arange(2 * 9 * 4 * 16, dtype=torch.float32).reshape(2, 9, 4, 16)
    bbox = torch.rand(2, 8, 3)
    za, ba = pipeline.align_lidar(z, bbox, 8)
    assert za.shape == (2, 9, 8, 8) and torch.equal(za[:, :, 2:6], z[..., 4:12]) and float(za[:, :, :2].abs().sum()) == 0
    assert torch.allclose(ba[..., 0], (bbox[..., 0] * 16 - 4) / 8) and torch.allclose(ba[..., 1], bbox[..., 1] + 0.25)


def test_postprocess_oracle_matches_reference_function():
    """oracle.postprocess against the reference's own inverse_depth_normalization / intensity expression
    (tests/golden/postprocess.npz): same torch expressions -> bit-exact, branch boundaries included."""
    from oracle import postprocess as opost
    g = load("postprocess")
    depth, inten = opost.range_denorm(g["sample"], g["min_d"], g["max_d"], alpha=float(g["alpha"]))
    assert torch.equal(depth, g["depth"])
    assert torch.equal(inten, g["intensity"])
    d0, i0 = opost.range_denorm(g["sample"], None, None, object_norm=False, int_norm=False)
    assert torch.equal(d0, g["sample"][:, [0]]) and torch.equal(i0, g["sample"][:, [1]])
