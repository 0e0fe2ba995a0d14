"""Dataset side on the device: `mobi_range_prepare` / `mobi_box_mask` against the host
path of `ldm.data.nuscenes.NuScenesDataset` (itself pinned to the reference's outputs in tests/test_data_side_cpu.py) and
against the transforms written out with numpy tiling / repeats; then a dataset batch through `get_input`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    from tests import mini_db
    root = str(tmp_path_factory.mktemp("mini_db_gpu"))
    return mini_db.build(root)


def _dataset(mini, **kw):
    from ldm.util import instantiate_from_config
    params = dict(state="test", use_lidar=True, use_camera=True, object_database_path=mini[0], scene_database_path=mini[1],
                  expand_mask_ratio=0.1, expand_ref_ratio=0, object_area_crop=0.2, num_samples_per_class=2, fixed_sampling=True,
                  object_random_crop=False, ref_aug=False, ref_mode="id-ref", image_height=128, image_width=128,
                  range_height=128, range_width=128, object_classes=["car", "pedestrian"], range_object_norm=True,
                  range_object_norm_scale=0.75, range_int_norm=True, min_lidar_points=8)
    params.update(kw)
    return instantiate_from_config({"target": "ldm.data.nuscenes.NuScenesDataset", "params": params})


def test_box_mask_matches_host_fill():
    from mobi_amd import ops
    from mobi_amd.ldm.data.utils import fill_box_faces
    rng = np.random.default_rng(3)
    H, W = 96, 160
    corners = []
    for k in range(10):
        c = np.array([rng.uniform(10, W - 10), rng.uniform(10, H - 10)])
        pts = c + rng.normal(0, 6 + 3 * k, (8, 2))
        corners.append(pts)
    corners.append(np.full((8, 2), 500.0))                       # off-image
    corners.append(np.array([[20.9, 20.2], [60.1, 20.7], [60.5, 50.5], [20.3, 50.9]] * 2))    # fractional corners: truncated
    corners = np.stack(corners)
    got, stats = ops.box_mask(torch.tensor(corners).cuda(), H, W, want_stats=True)
    got, stats = got.cpu().numpy(), stats.cpu().numpy()
    for k in range(len(corners)):
        filled = fill_box_faces(corners[k], H, W) > 0.5
        assert np.array_equal(got[k], (1.0 - filled).astype(np.float32)), k
        ys, xs = np.nonzero(filled)
        want = [filled.sum(), xs.min(), xs.max(), ys.min(), ys.max()] if filled.any() else [0, W, -1, H, -1]
        assert list(stats[k]) == want, k
    assert got[-2].min() == 1 and got[-1][20:51, 20:61].max() == 0 and got[-1].sum() == H * W - 31 * 41
    only = ops.box_mask(torch.tensor(corners).cuda(), H, W, want_mask=False, want_stats=True)
    assert torch.equal(only.cpu(), torch.from_numpy(stats))


@pytest.mark.parametrize("int_norm", [False, True])
def test_range_prepare_matches_written_out_transforms(int_norm):
    """Power-of-two windows incl. one that wraps around the sweep, against np.tile / slice / np.repeat and the dataset's
    torch expressions; a 96-column window (general nearest rule) against LidarConverter._nearest."""
    from mobi_amd import ops
    from mobi_amd.ldm.data.lidar_converter import LidarConverter
    from mobi_amd.ldm.data.utils import depth_normalization
    rng = np.random.default_rng(5)
    B, H0, W0, R = 5, 32, 1096, 256
    depth = rng.uniform(-1, 1, (B, H0, W0)).astype(np.float32)
    depth[:, :, ::7] = -1.0                                            # empty pixels
    inten = rng.integers(0, 256, (B, H0, W0)).astype(np.float32)
    inst = (rng.uniform(0, 1, (B, H0, W0)) > 0.9).astype(np.float32)
    crop_left = np.array([1096 + 100, 2 * 1096 - 40, 1096 - 30, 1500, 1096 + 1000])
    width_crop = np.array([64, 128, 256, 96, 256])
    lo = np.array([-0.6, -0.2, 0.1, -0.9, -0.5], dtype=np.float32)
    hi = np.array([-0.3, 0.3, 0.4, 0.9, -0.45], dtype=np.float32)
    mask = (rng.uniform(0, 1, (B, 1, R, R)) > 0.3).astype(np.float32)
    t = lambda a: torch.tensor(a).cuda()
    rd, rdi, io = ops.range_prepare(t(depth), t(inten), t(inst), t(crop_left), t(width_crop), t(lo), t(hi), t(mask), height=R,
                                    width=R, alpha=0.75, object_norm=True, int_norm=int_norm)
    for b in range(B):
        views = []
        for a in (depth[b], inten[b], inst[b]):
            win = np.tile(a, 3)[:, crop_left[b]:crop_left[b] + width_crop[b]]
            if R % width_crop[b] == 0:
                views.append(np.repeat(np.repeat(win, R // H0, 0), R // width_crop[b], 1))
            else:
                views.append(LidarConverter._nearest(win, R, R))
        d = depth_normalization(torch.from_numpy(views[0])[None], torch.tensor(lo[b]), torch.tensor(hi[b]), alpha=0.75)
        v = torch.from_numpy(((views[1] / 255) - 0.5) * 2)[None]
        if int_norm:
            v = torch.clamp(2 * (1 - torch.exp(-2 * (v + 1))) - 1, -1, 1)
        assert torch.equal(rd[b, :1].cpu(), d), b                        # piecewise-linear map: bit for bit
        if int_norm:
            assert float((rd[b, 1:].cpu() - v).abs().max()) <= 2e-7, b   # expf on the device vs torch's exp
        else:
            assert torch.equal(rd[b, 1:].cpu(), v), b
        assert torch.equal(rdi[b].cpu(), rd[b].cpu() * torch.from_numpy(mask[b])), b
        assert torch.equal(io[b, 0].cpu(), torch.from_numpy(views[2])), b
    from mobi_amd import _lib
    with pytest.raises(ValueError):
        ops.range_prepare(t(depth), t(inten), t(inst), t(crop_left), width_crop * 8, t(lo), t(hi), t(mask), height=R, width=R)
    with pytest.raises(_lib.EngineError):                              # a REDUCING view is the reference's pooling branch
        ops.range_prepare(t(depth), t(inten), None, t(crop_left), t(np.minimum(width_crop, 16)), t(lo), t(hi), t(mask[:, :, :16, :16].copy()),
                          height=16, width=16)


@pytest.mark.parametrize("int_norm", [False, True])
def test_collate_device_equals_host_items(mini, int_norm):
    from torch.utils.data import default_collate
    np.random.seed(0)
    ds = _dataset(mini, range_int_norm=int_norm)
    host = default_collate([ds[i] for i in range(len(ds))])
    dev = ds.collate_device([ds.raw_item(i) for i in range(len(ds))])
    hl, dl = host["lidar"], dev["lidar"]
    assert torch.equal(dl["range_mask"].cpu(), hl["range_mask"])
    assert torch.equal(dl["range_data"][:, :1].cpu(), hl["range_data"][:, :1])
    tol = 2e-7 if int_norm else 0.0
    assert float((dl["range_data"][:, 1:].cpu() - hl["range_data"][:, 1:]).abs().max()) <= tol
    assert float((dl["range_data_inpaint"].cpu() - hl["range_data_inpaint"]).abs().max()) <= tol
    assert torch.equal(dl["range_instance_mask"].cpu(), hl["range_instance_mask"])
    # camera: frame normalisation, edit mask, crop and the bilinear resize in one launch; torch's CPU bilinear and the
    # kernel evaluate the same four-tap expression (rounding may differ in the last bit of a product)
    for k in ("GT", "inpaint_image", "inpaint_mask"):
        assert float((dev["image"][k].cpu() - host["image"][k]).abs().max()) <= 1e-6, k
    assert torch.equal(dev["image"]["cond"]["ref_bbox"].cpu(), host["image"]["cond"]["ref_bbox"])
    assert torch.equal(dev["image"]["cond"]["ref_image"].cpu(), host["image"]["cond"]["ref_image"])
    assert torch.equal(dl["cond"]["ref_bbox"].cpu(), hl["cond"]["ref_bbox"]) and dev["id_name"] == host["id_name"]
    assert "range_mask_corners" not in dl and dl["range_data"].is_cuda and "frame" not in dev["image"] and "erase" not in dev
    assert set(dev["image"]) == set(host["image"]) and set(dl) == set(hl)
    if int_norm:
        dso = _dataset(mini, return_original_image=True, object_classes=["car"])
        h = default_collate([dso[i] for i in range(len(dso))])["image"]["orig"]
        d = dso.collate_device([dso.raw_item(i) for i in range(len(dso))])["image"]["orig"]
        assert torch.equal(d["crop"].cpu(), h["crop"]) and torch.equal(d["mask"].cpu(), h["mask"])
        assert float((d["image"].cpu() - h["image"]).abs().max()) <= 2.4e-7 and d["file_name"] == h["file_name"]   # x / 255: 1 ulp
        cam_only = _dataset(mini, use_lidar=False)
        hc = default_collate([cam_only[i] for i in range(2)])
        dc = cam_only.collate_device([cam_only.raw_item(i) for i in range(2)])
        assert torch.equal(dc["image"]["cond"]["ref_bbox"].cpu(), hc["image"]["cond"]["ref_bbox"]) and dc["lidar"] == {}


def test_dataset_batch_feeds_get_input(mini):
    """A DataLoader batch of the dataset through `LatentDiffusion.get_input` (reduced widths), as the harness does
    (scripts/inference_test_bench.py:383-416)."""
    import mobi_amd
    from mobi_amd.ldm.util import instantiate_from_config, load_config
    from oracle import weights as W
    import os
    mobi_amd.set_engine_dtype(torch.float16)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "configs", "mobi_nusc_256.yaml"),
                      ["latent_size=16", "image_height=128", "use_lidar=True",
                       "model.params.lidar_stage_config.params.ckpt_path=null",
                       "model.params.unet_config.params.model_channels=64",
                       "model.params.first_stage_config.params.ddconfig.ch=32",
                       "model.params.lidar_stage_config.params.ddconfig.ch=32",
                       "model.params.cond_stage_config.params.clip_config.hidden_size=1024",
                       "model.params.cond_stage_config.params.clip_config.intermediate_size=256",
                       "model.params.cond_stage_config.params.clip_config.num_hidden_layers=1",
                       "model.params.cond_stage_config.params.clip_config.num_attention_heads=16"])
    model = instantiate_from_config(cfg["model"])
    W.fill_module_(model, seed=23)
    model = model.cuda().eval()
    ds = _dataset(mini)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0, shuffle=False)))
    move = lambda d: {k: move(v) if isinstance(v, dict) else (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    batch = move(batch)
    with torch.no_grad():
        data = model.get_input(batch, model.first_stage_key, force_c_encode=True, return_vae_rec=True)
    assert data["z"].shape == (4, 9, 16, 16) and data["cond"].shape == (4, 2, 768) and data["z_lidar"].shape == (2, 4, 16, 16)
    assert torch.isfinite(data["z"]).all() and torch.isfinite(data["cond"]).all()
    assert data["image_rec"].shape == (2, 3, 128, 128) and data["lidar_rec"].shape == (2, 2, 128, 128)


def test_harness_loop_on_mini_db(mini):
    """scripts/inference_test_bench.py:376-610 end to end on the miniature database, every stage on the engine:
    dataset (return_original_image, as `--save_samples` sets it) -> DataLoader -> get_input -> DDIM -> decode_sample ->
    log_data -> camera paste-back into the full frame (orig.crop / orig.mask) and range-view paste into the full sweep
    (range_shift_left / width_crop / pitch / yaw / bbox_3d / instance mask)."""
    import os
    import mobi_amd
    from ldm.util import instantiate_from_config
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.data import utils as du
    from mobi_amd.ldm.util import load_config
    from oracle import weights as W
    mobi_amd.set_engine_dtype(torch.float16)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "configs", "mobi_nusc_256.yaml"),
                      ["latent_size=16", "image_height=128", "use_lidar=True",
                       "model.params.lidar_stage_config.params.ckpt_path=null",
                       "model.params.unet_config.params.model_channels=64",
                       "model.params.first_stage_config.params.ddconfig.ch=32",
                       "model.params.lidar_stage_config.params.ddconfig.ch=32",
                       "model.params.cond_stage_config.params.clip_config.hidden_size=1024",
                       "model.params.cond_stage_config.params.clip_config.intermediate_size=256",
                       "model.params.cond_stage_config.params.clip_config.num_hidden_layers=1",
                       "model.params.cond_stage_config.params.clip_config.num_attention_heads=16"])
    model = instantiate_from_config(cfg["model"])
    W.fill_module_(model, seed=29)
    model = model.cuda().eval()
    sampler = DDIMSampler(model)
    ds = _dataset(mini, return_original_image=True)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0, pin_memory=True, shuffle=False, drop_last=False)
    move = lambda d: {k: move(v) if isinstance(v, dict) else (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    seen = 0
    with torch.no_grad(), model.ema_scope():
        for batch in loader:
            ids = batch["id_name"]
            batch = move(batch)
            data = model.get_input(batch, model.first_stage_key, force_c_encode=True, return_vae_rec=True)
            n = data["z"].shape[0]
            uc = torch.cat([model.learnable_vector.repeat(n, 1, 1), model.bbox_uncond_vector.repeat(n, 1, 1)], dim=1)
            shape = [model.channels, model.image_size, model.image_size]
            samples, _ = sampler.sample(S=4, conditioning=data["cond"], batch_size=n, shape=shape, verbose=False,
                                        unconditional_guidance_scale=5.0, unconditional_conditioning=uc, eta=0.0,
                                        x_T=torch.randn([n, *shape], device="cuda"),
                                        test_model_kwargs={"inpaint_image": data["z"][:, 4:8], "inpaint_mask": data["z"][:, [8]]})
            h_cam, h_lid = model.decode_sample(samples, data.get("z_lidar"))
            log, metrics = model.log_data(batch, data, h_cam, h_lid, log_metrics=False, return_sample=True, split="test")
            assert metrics is not None and all(np.isfinite(v) or np.isnan(v) for v in metrics.values())
            B = len(ids)
            for i in range(B):                                        # camera: :478-510
                orig = batch["image"]["orig"]
                recon, pred = du.paste_camera_patch(patch_pred=log["image_sample"][[i]], image=orig["image"][i],
                                                    mask=orig["mask"][i], crop=orig["crop"][i])
                assert recon.shape == (450, 800, 3) and pred.dtype == torch.uint8 and torch.isfinite(recon).all()
                left, top, cw, ch = (int(v) for v in orig["crop"][i])
                assert int(pred[top:top + ch, left:left + cw].sum()) > 0 and int(pred.sum()) == int(pred[top:top + ch, left:left + cw].sum())
                far = (orig["mask"][i] == 1)                           # far from the edit region the frame is untouched
                far[max(0, top - 30):top + ch + 30, max(0, left - 30):left + cw + 30] = False
                frame_bgr = ((orig["image"][i].permute(1, 2, 0).flip(-1) + 1) / 2 * 255).to(torch.uint8).float()
                assert float((recon[far] - frame_bgr[far]).abs().max()) <= 1.0
            lid = batch["lidar"]                                       # lidar: :567-610, the whole batch in one launch
            out = du.paste_range_objects(range_depth=log["range_sample_depth"], range_int=log["range_sample_int"],
                                         range_depth_orig=lid["range_depth_orig"], range_int_orig=lid["range_int_orig"],
                                         crop_left=lid["range_shift_left"], width_crop=lid["width_crop"],
                                         range_pitch=lid["range_pitch"], range_yaw=lid["range_yaw"], bbox_3d=batch["bbox_3d"],
                                         gt_instance_mask=lid["range_instance_mask_orig"])
            assert out["depth_final"].shape == (B, 32, 1096) and torch.isfinite(out["depth_final"]).all()
            changed = (out["depth_final"] != lid["range_depth_orig"].float())
            inside = (out["pred_mask"] != 0) | (lid["range_instance_mask_orig"] != 0)
            assert not (changed & ~inside).any()                       # the sweep only changes where an object is / was
            seen += B
    assert seen == len(ds)
