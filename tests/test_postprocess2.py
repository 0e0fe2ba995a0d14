"""Harness post-processing, second slice: range-view un-crop + points-in-box paste, per-sample
lidar error scores, camera paste-back.

CPU part: the oracle (oracle/postprocess.py) against tests/golden/postprocess2.npz, which the REFERENCE's functions
produced (postprocess_range_depth_int, range2pcd, points_in_bbox_corners, LatentDiffusion.log_data).
GPU part: the engine (mobi_range_paste, mobi_lidar_metrics, mobi_paste_patch / mobi_gaussian_blur / mobi_blend_frame
through the `ldm.data.*` mirrors and LatentDiffusion.log_data) against the same golden -- un-crop, paste and median scores
BIT-EXACT, RMSE scores to 1e-6 relative (torch's fp32 mean has its own summation order) -- and against the oracle for
the camera paste-back, whose reference is cv2 (absent here: "parity unpinned", tolerances stated in the test)."""
import numpy as np
import pytest
import torch

from oracle import postprocess as op, weights as W
from tests.golden_cases import load


def _np(t):
    return t.numpy() if isinstance(t, torch.Tensor) else t


def test_oracle_uncrop_and_paste_match_the_reference():
    g = load("postprocess2")
    for i in range(4):
        d = op.undo_default_transforms(int(g["unc_crop_left"][i]), int(g["unc_width_crop"][i]), _np(g["unc_depth"][i, 0]),
                                       _np(g["unc_d_orig"][i]))
        it = op.undo_default_transforms(int(g["unc_crop_left"][i]), int(g["unc_width_crop"][i]), _np(g["unc_int"][i, 0]),
                                        _np(g["unc_i_orig"][i]))
        assert np.array_equal(d, _np(g["unc_depth_out"][i])) and np.array_equal(it, _np(g["unc_int_out"][i]))
        pm, df, itf = op.paste_object(d, it, _np(g["unc_d_orig"][i]), _np(g["unc_i_orig"][i]), _np(g["paste_pitch"][i]),
                                      _np(g["paste_yaw"][i]), _np(g["paste_boxes"][i]), _np(g["paste_gt_mask"][i]))
        assert pm.sum() > 0 and np.array_equal(pm, _np(g["paste_pred_mask"][i]) != 0)
        assert np.array_equal(df, _np(g["paste_depth_final"][i])) and np.array_equal(itf, _np(g["paste_int_final"][i]))


def _reference_scores(g):
    return dict(zip([str(k) for k in g["met_keys"]], _np(g["met_values"])))


def _oracle_metric_inputs(g):
    alpha = 0.75
    den = lambda t: op.range_denorm(t, g["met_min_d"], g["met_max_d"], alpha=alpha, object_norm=True, int_norm=True)
    sd, si = den(g["met_sample"])
    rd, ri = den(g["met_rec"])
    idp, ii = den(g["met_in"])
    return {"pred_depth": (sd, idp), "rec_depth": (rd, idp), "pred_int": (si, ii), "rec_int": (ri, ii)}


def test_oracle_lidar_scores_match_the_reference():
    g = load("postprocess2")
    ref = _reference_scores(g)
    box = 1 - g["met_rmask"]
    for name, (p, q) in _oracle_metric_inputs(g).items():
        sc = op.lidar_scores(p, q, g["met_inst"], box, g["met_width_crop"])
        scale = (54 - 1.4) / 2 if "depth" in name else 128
        for si, score in ((0, "mse"), (1, "median_error")):
            obj = sc[:, 0, si]
            obj = obj[~np.isnan(obj)]
            assert len(obj) == 2                                    # sample 2 has no object pixels: dropped, as the reference
            assert np.isclose(obj.mean() * scale, ref[f"test/{score}/object_{name}"], rtol=1e-12)
            assert np.isclose(sc[:, 1, si].mean() * scale, ref[f"test/{score}/mask_{name}"], rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
def test_gpu_uncrop_and_paste_bit_exact():
    from mobi_amd.ldm.data import utils as du
    g = load("postprocess2")
    c = lambda t: t.cuda()
    d, it = du.postprocess_range_depth_int(range_depth=c(g["unc_depth"]), range_depth_orig=c(g["unc_d_orig"]),
                                           range_int=c(g["unc_int"]), range_int_orig=c(g["unc_i_orig"]),
                                           crop_left=g["unc_crop_left"], width_crop=g["unc_width_crop"])
    assert isinstance(d, np.ndarray) and np.array_equal(d, _np(g["unc_depth_out"])) and np.array_equal(it, _np(g["unc_int_out"]))
    out = du.paste_range_objects(range_depth=c(g["unc_depth"]), range_int=c(g["unc_int"]), range_depth_orig=c(g["unc_d_orig"]),
                                 range_int_orig=c(g["unc_i_orig"]), crop_left=g["unc_crop_left"],
                                 width_crop=g["unc_width_crop"], range_pitch=g["paste_pitch"], range_yaw=g["paste_yaw"],
                                 bbox_3d=g["paste_boxes"], gt_instance_mask=g["paste_gt_mask"])
    assert torch.equal(out["depth_unc"].cpu(), g["unc_depth_out"])
    assert np.array_equal(out["pred_mask"].cpu().numpy() != 0, _np(g["paste_pred_mask"]) != 0)
    assert torch.equal(out["depth_final"].cpu(), g["paste_depth_final"]) and torch.equal(out["int_final"].cpu(), g["paste_int_final"])
    # the stand-alone mirrors
    from mobi_amd.ldm.data.box_np_ops import points_in_bbox_corners
    from mobi_amd.ldm.data.lidar_converter import LidarConverter
    conv = LidarConverter(H=8, W=137)
    pts, lab, beam = conv.range2pcd(d[2], _np(g["paste_pitch"][2]), _np(g["paste_yaw"][2]), np.arange(8 * 137).reshape(8, 137))
    inside = points_in_bbox_corners(pts, _np(g["paste_boxes"][[2]]))
    m = np.zeros(8 * 137)
    m[lab[inside[:, 0]]] = 1
    assert np.array_equal(m.reshape(8, 137) != 0, _np(g["paste_pred_mask"][2]) != 0) and beam.shape == lab.shape
    d1, _ = conv.undo_default_transforms(int(g["unc_crop_left"][1]), int(g["unc_width_crop"][1]), _np(g["unc_depth"][1, 0]),
                                         _np(g["unc_d_orig"][1]))
    assert np.array_equal(d1, _np(g["unc_depth_out"][1]))


@gpu
def test_gpu_log_data_scores_and_keys():
    """LatentDiffusion.log_data on the engine: the reference's metric dict, range_sample_depth and collage layout."""
    import mobi_amd
    from mobi_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    mobi_amd.set_engine_dtype(torch.float16)
    g = load("postprocess2")
    ref = _reference_scores(g)

    class Stub(LatentDiffusion):
        def __init__(self):                                       # no networks: decode_first_stage is given
            torch.nn.Module.__init__(self)
            self.use_camera, self.use_lidar = False, True
            self.range_object_norm, self.range_object_norm_scale, self.range_int_norm = True, 0.75, True

        def decode_first_stage(self, z, **kw):
            return g["met_sample"].cuda()

    c = lambda t: t.cuda()
    batch = {"lidar": {"range_data": c(g["met_in"]), "range_data_inpaint": c(g["met_in"] * g["met_rmask"]),
                       "range_mask": c(g["met_rmask"]), "range_instance_mask": c(g["met_inst"]),
                       "min_depth_obj": c(g["met_min_d"]), "max_depth_obj": c(g["met_max_d"]),
                       "width_crop": g["met_width_crop"]}}
    log, metrics = Stub().log_data(batch, {"lidar_rec": c(g["met_rec"])}, None, None, log_metrics=False, return_sample=True,
                                   split="test")
    assert sorted(metrics) == sorted(ref)
    for k, v in ref.items():
        # depth scores: the de-normalisation is bit-exact, so medians (selections) are exact and RMSEs differ only by the
        # summation order of torch's fp32 mean; intensity scores go through a logarithm (device logf vs torch-CPU log)
        tol = 1e-5 if "int" in k else (0 if "median" in k else 1e-6)
        assert abs(metrics[k] - v) <= tol * abs(v), (k, metrics[k], v)
    assert torch.equal(log["range_sample_depth"].cpu(), g["met_range_sample_depth"])
    assert tuple(log["range_depth_pred"].shape) == tuple(int(v) for v in g["met_depth_pred_rows"])
    assert abs(float(log["range_depth_pred"].double().sum()) - float(g["met_depth_pred_sum"])) < 1e-6
    assert abs(float(log["range_int_pred"].double().sum()) - float(g["met_int_pred_sum"])) < 1e-6


@gpu
def test_gpu_camera_paste_back_vs_oracle():
    """Unpinned against cv2 (absent): engine vs the oracle's restatement.  Blur and blend to 2e-6 absolute on a 0..255
    scale; the resized uint8 patch within 1 LSB, on at most 0.5 % of its bytes (F.interpolate's CPU kernel may fuse its
    multiply-adds differently)."""
    from mobi_amd.ldm.data import utils as du
    H, Wd = 90, 160
    patch = torch.clamp(W.synth_input("pb.patch", (1, 3, 64, 64)) * 0.5, -1, 1)
    image = torch.clamp(W.synth_input("pb.image", (3, H, Wd)) * 0.5, -1, 1)
    mask = torch.ones(H, Wd)
    mask[20:70, 40:130] = 0
    crop = (30, 10, 101, 77)                                       # left, top, crop_W, crop_H: odd sizes, off-centre
    ref_recon, ref_pred = op.paste_camera_patch(patch, image, mask, crop)
    recon, pred = du.paste_camera_patch(patch_pred=patch.cuda(), image=image.cuda(), mask=mask.cuda(), crop=crop)
    diff = np.abs(pred.cpu().numpy().astype(np.int32) - ref_pred.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 5e-3
    blur = __import__("mobi_amd").ops.gaussian_blur(mask.cuda(), torch.from_numpy(op.gaussian_kernel1d(15, 7.0)).cuda())
    assert np.abs(blur.cpu().numpy() - op.gaussian_blur_reflect101(mask.numpy(), 15, 7.0)).max() < 2e-6
    same = diff.max(-1) == 0                                       # compare the blend where the uint8 patch agrees
    assert np.abs(recon.cpu().numpy() - ref_recon)[same].max() < 2e-3
