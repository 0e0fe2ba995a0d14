#!/usr/bin/env python3
"""tests/golden/postprocess2.npz: harness post-processing produced by the REFERENCE's own functions on CPU
(build container only):
  * ldm.data.utils.postprocess_range_depth_int (-> LidarConverter.undo_default_transforms, pool_resize) on range samples
    whose crop windows include one that wraps around the sweep;
  * the range-view paste of scripts/inference_test_bench.py:583-610 -- that code is inline in the harness's main(), so its
    few statements are restated here AROUND the reference's functions (LidarConverter.range2pcd,
    ldm.data.box_np_ops.points_in_bbox_corners with numba's decorators reduced to plain Python);
  * LatentDiffusion.log_data's per-sample lidar error scores (ddpm.py:1545-1597), by calling the reference's log_data on a
    stand-in object (decode_first_stage returns the given sample, the point-cloud pictures are skipped).
Reduced geometry (64 x 64 samples, 8 x 137 sweep for the paste, 32-row pooling for the scores) -- the code paths are the
full-size ones.   python tests/golden/make_golden_postprocess2.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402
from oracle import weights as W                           # noqa: E402


def main():
    numba = types.ModuleType("numba")
    deco = lambda *a, **k: (a[0] if a and callable(a[0]) and not k else (lambda f: f))
    numba.jit = numba.njit = deco
    sys.modules["numba"] = numba
    R = mg.import_reference()
    import ldm.data.utils as du
    import ldm.data.box_np_ops as bo
    from ldm.data.lidar_converter import LidarConverter
    out = {}

    # ---- 1. un-crop (postprocess_range_depth_int) ------------------------------------------------------------------
    B, hc, wc, h0, w0 = 4, 64, 64, 8, 137
    u = lambda n, s: torch.clamp(W.synth_input(n, s) * 0.6, -1, 1)
    depth, inten = u("pp2.depth", (B, 1, hc, wc)), u("pp2.int", (B, 1, hc, wc))
    d_orig, i_orig = u("pp2.d0", (B, h0, w0)), u("pp2.i0", (B, h0, w0))
    crop_left = torch.tensor([5, 120, 137 + 40, 0])                 # 120 + 32 > 137: wraps; 177 % 137 = 40
    width_crop = torch.tensor([16, 32, 64, 64])
    d_unc, i_unc = du.postprocess_range_depth_int(range_depth=depth, range_depth_orig=d_orig, range_int=inten,
                                                  range_int_orig=i_orig, crop_left=crop_left, width_crop=width_crop)
    out.update(unc_depth=depth, unc_int=inten, unc_d_orig=d_orig, unc_i_orig=i_orig, unc_crop_left=crop_left,
               unc_width_crop=width_crop, unc_depth_out=d_unc, unc_int_out=i_unc)

    # ---- 2. paste (inference_test_bench.py:583-610 around the reference's range2pcd / points_in_bbox_corners) --------
    conv = LidarConverter(H=h0, W=w0)
    yaw = np.tile(np.linspace(np.pi, -np.pi, w0, dtype=np.float32)[None], (h0, 1))
    pitch = np.tile(np.linspace(0.18, -0.5, h0, dtype=np.float32)[:, None], (1, w0))
    yaw, pitch = np.stack([yaw] * B), np.stack([pitch] * B)
    # boxes that catch a patch of the sweep: built around a point of the un-cropped prediction
    boxes, finals_d, finals_i, pred_masks = [], [], [], []
    gt_mask = (W.synth_input("pp2.gt", (B, h0, w0)) > 1.2).numpy()
    for i in range(B):
        label = np.arange(0, h0 * w0).reshape(h0, w0)
        points, points_label, _ = conv.range2pcd(d_unc[i], pitch[i], yaw[i], label)
        c = points[len(points) // 3]
        half = np.array([4.0, 3.0, 2.5], dtype=np.float32) * (1 + 0.3 * i)
        sx = np.array([[-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1]],
                      dtype=np.float32)
        th = 0.4 * i
        rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=np.float32)
        box = ((sx * half) @ rot.T + c).astype(np.float32)[None]       # [1, 8, 3], corner order of the nuScenes boxes
        object_points = bo.points_in_bbox_corners(points, box)
        pred = np.zeros(h0 * w0)
        pred[points_label[object_points[:, 0]]] = 1
        pred = pred.reshape(h0, w0)
        inst = np.logical_or(pred, gt_mask[i])
        finals_d.append(np.where(inst, d_unc[i], d_orig[i].numpy()))
        finals_i.append(np.where(inst, i_unc[i], i_orig[i].numpy()))
        pred_masks.append(pred)
        boxes.append(box[0])
    assert sum(m.sum() for m in pred_masks) > 20, "boxes caught nothing: the golden would be vacuous"
    out.update(paste_pitch=pitch, paste_yaw=yaw, paste_boxes=np.stack(boxes), paste_gt_mask=gt_mask,
               paste_pred_mask=np.stack(pred_masks), paste_depth_final=np.stack(finals_d), paste_int_final=np.stack(finals_i))

    # ---- 3. lidar error scores (LatentDiffusion.log_data, ddpm.py:1499-1597) ------------------------------------------
    LD = R.ddpm.LatentDiffusion
    Bm, H = 3, 64
    sample = u("pp2.m.sample", (Bm, 2, H, H))
    rec = u("pp2.m.rec", (Bm, 2, H, H))
    data_in = u("pp2.m.in", (Bm, 2, H, H))
    inst = (W.synth_input("pp2.m.inst", (Bm, 1, H, H)) > 1.0).float()
    inst[2] = 0                                                       # a sample without object pixels: NaN -> dropped
    rmask = (W.synth_input("pp2.m.mask", (Bm, 1, H, H)) < 0.8).float()
    min_d, max_d = torch.tensor([-0.6, -0.9, 0.1]), torch.tensor([0.3, -0.2, 0.95])
    wcrop = torch.tensor([16, 32, 64])

    class Fake:
        use_camera, use_lidar = False, True
        range_object_norm, range_object_norm_scale, range_int_norm = True, 0.75, True
        decode_first_stage = lambda self, h, module_name=None: sample.clone()
        log_dict = lambda self, *a, **k: None

    R.ddpm.get_lidar_vis = lambda **k: (torch.zeros(Bm, 3, 4, 4),) * 3
    batch = {"lidar": {"range_data": data_in.clone(), "range_data_inpaint": data_in * rmask, "range_mask": rmask,
                       "range_instance_mask": inst, "min_depth_obj": min_d, "max_depth_obj": max_d, "width_crop": wcrop,
                       "range_depth_orig": None, "range_shift_left": None, "range_pitch": None, "range_yaw": None},
             "bbox_3d": None}
    log, metrics = LD.log_data(Fake(), batch, {"lidar_rec": rec.clone()}, None, None, log_metrics=False,
                               return_sample=True, split="test")
    keys = sorted(metrics)
    out.update(met_sample=sample, met_rec=rec, met_in=data_in, met_inst=inst, met_rmask=rmask, met_min_d=min_d,
               met_max_d=max_d, met_width_crop=wcrop, met_keys=np.array(keys), met_values=np.array([metrics[k] for k in keys]),
               met_range_sample_depth=log["range_sample_depth"],
               # the collages are concatenations of tensors stored above: only their layout is recorded
               met_depth_pred_rows=np.array(log["range_depth_pred"].shape),
               met_depth_pred_sum=log["range_depth_pred"].double().sum(), met_int_pred_sum=log["range_int_pred"].double().sum())
    mg.save("postprocess2", **out)


if __name__ == "__main__":
    main()
