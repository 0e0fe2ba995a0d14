#!/usr/bin/env python3
"""tests/golden/data_side.npz: the dataset-side geometry produced by the REFERENCE's own functions on CPU (build
container only).  Only what runs WITHOUT cv2 / torchvision / albumentations (absent from the image; the stubs that let
`ldm.data.utils` import are empty) is generated here:
  * LidarConverter.pcd2range on a synthetic sweep (no two points at the same depth in one pixel: the reference's
    `np.argsort(depth)[::-1]` is not stable, ties are undefined there), get_range_coords, tile, bbox_crop, the pooling
    branch of resize, and apply_default_transforms on box coordinates alone (the call get_range_inpaint_mask makes);
  * ldm.data.utils.get_image_coords / rotate_bbox / translate_bbox / get_camera_coords / expand_bbox_corners /
    get_2d_bbox, get_inpaint_mask with use_3d_edit_mask=False.
cv2.resize(INTER_NEAREST) and cv2.fillPoly are restated from OpenCV's documentation in oracle/data_side.py and are NOT
pinned by this file.   python tests/golden/make_golden_data_side.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402


def boxes(rng, n):
    """[n, 8, 3] box corners in the nuScenes corner order, 4-45 m from the sensor."""
    sx = np.array([[-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1]], dtype=np.float64)
    out = []
    for i in range(n):
        half = np.array([rng.uniform(0.3, 2.5), rng.uniform(0.3, 1.2), rng.uniform(0.4, 1.0)])
        th, r, az = rng.uniform(-np.pi, np.pi), rng.uniform(4, 45), rng.uniform(-np.pi, np.pi)
        rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        c = np.array([r * np.cos(az), r * np.sin(az), rng.uniform(-1.5, 0.5)])
        out.append((sx * half) @ rot.T + c)
    return np.stack(out)


def main():
    mg.import_reference()
    import ldm.data.utils as du
    from ldm.data.lidar_converter import LidarConverter
    rng = np.random.default_rng(20240607)
    out = {}

    # ---- pcd2range ---------------------------------------------------------------------------------------------------
    n = 30000
    az = rng.uniform(-np.pi, np.pi, n)
    beam = rng.integers(0, 32, n)
    pitch = 0.0232 * (beam - 23) + rng.normal(0, 0.002, n)
    dist = rng.uniform(0.5, 70.0, n)                               # some outside (1.4, 54): filtered
    pts = np.stack([dist * np.cos(pitch) * np.cos(az), dist * np.cos(pitch) * np.sin(az), dist * np.sin(pitch)], 1).astype(np.float32)
    label = rng.integers(0, 256, n).astype(np.float32)
    for tag, log_scale in (("lin", False), ("log", True)):
        conv = LidarConverter(log_scale=log_scale)
        d, i, keep, p, y = conv.pcd2range(pts, label)
        out.update({f"p2r_{tag}_depth": d, f"p2r_{tag}_int": i, f"p2r_{tag}_keep": keep, f"p2r_{tag}_pitch": p, f"p2r_{tag}_yaw": y})
    out.update(p2r_points=pts, p2r_label=label)

    # ---- box geometry ------------------------------------------------------------------------------------------------
    bx = boxes(rng, 12)
    out["boxes"] = bx
    out["range_coords"] = np.stack([LidarConverter().get_range_coords(b) for b in bx])
    out["range_coords_log"] = np.stack([LidarConverter(log_scale=True).get_range_coords(b) for b in bx])
    # the coordinate-only pass of apply_default_transforms (get_range_inpaint_mask, data/utils.py:174-198), with the
    # window derived from the box and with a given window
    adt, adt_given = [], []
    for k, b in enumerate(bx):
        conv = LidarConverter()
        c = conv.get_range_coords(du.expand_bbox_corners(b, 0.1))
        _, _, _, c2, crop_left, width_crop = conv.apply_default_transforms(c, height=512, width=512)
        adt.append(np.concatenate([c2.reshape(-1), [crop_left, width_crop]]))
        conv = LidarConverter()
        c = conv.get_range_coords(du.expand_bbox_corners(b, 0.1))
        _, _, _, c3, cl3, wc3 = conv.apply_default_transforms(c, height=256, width=256, crop_left=1096 + 37 * k, width_crop=128)
        adt_given.append(np.concatenate([c3.reshape(-1), [cl3, wc3]]))
    out["adt_coords"], out["adt_given"] = np.stack(adt), np.stack(adt_given)

    # tile + bbox_crop + pooling resize on arrays (no cv2 on this path: the crop is resized DOWN by whole factors)
    conv = LidarConverter()
    depth = rng.uniform(-1, 1, (32, 1096)).astype(np.float32)
    inten = rng.uniform(0, 255, (32, 1096)).astype(np.float32)
    mask = (rng.uniform(0, 1, (32, 1096)) > 0.9).astype(np.float32)
    c = conv.get_range_coords(bx[0])
    d3, i3, m3, c3 = conv.tile(depth, inten, mask, c, n=3)
    d4, i4, m4, c4, crop_left = conv.bbox_crop(c3, d3, i3, m3, width=256)
    d5, i5, m5, c5 = conv.resize(d4, i4, m4, c4, new_W=64, new_H=16)
    out.update(rv_depth=depth, rv_int=inten, rv_mask=mask, rv_crop_left=np.int64(crop_left), rv_crop_depth=d4, rv_crop_int=i4,
               rv_crop_mask=m4, rv_crop_coords=c4, rv_pool_depth=d5, rv_pool_int=i5, rv_pool_mask=m5, rv_pool_coords=c5)

    # ---- camera-side geometry ----------------------------------------------------------------------------------------
    K = np.array([[1266.4, 0, 816.3, 0], [0, 1266.4, 491.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    l2c = np.array([[0, -1, 0, 0.1], [0, 0, -1, -0.3], [1, 0, 0, -0.5], [0, 0, 0, 1]], dtype=np.float64)     # x forward -> camera z
    l2i = K @ l2c
    fwd = bx.copy()
    fwd[..., 0] = np.abs(fwd[..., 0]) + 3.0                                       # in front of the camera
    out.update(cam_boxes=fwd, lidar2image=l2i, lidar2camera=l2c,
               image_coords=np.stack([du.get_image_coords(b, l2i, include_depth=True) for b in fwd]),
               image_coords_2d=np.stack([du.get_image_coords(b, l2i) for b in fwd]),
               camera_coords=np.stack([du.get_camera_coords(b, l2c) for b in fwd]),
               rotated=np.stack([du.rotate_bbox(b.copy(), 30.0 * k) for k, b in enumerate(fwd)]),
               translated=np.stack([du.translate_bbox(b.copy(), np.array([3.0, -10.0, -1.5])) for b in fwd]),
               expanded=np.stack([du.expand_bbox_corners(b.copy(), 0.1) for b in fwd]),
               bbox_2d=np.stack([du.get_2d_bbox(b.copy(), l2i, 900, 1600, 0.1) for b in fwd]),
               mask_2d=np.stack([du.get_inpaint_mask(b.copy(), l2i, 90, 160, 0.1, use_3d_edit_mask=False).numpy()
                                 for b in (fwd[:4] * np.array([1.0, 1.0, 1.0]))]))
    # (the 90 x 160 masks use the same projection scaled by 0.1: boxes divided accordingly)
    l2i_small = np.diag([0.1, 0.1, 1.0, 1.0]) @ l2i
    out["lidar2image_small"] = l2i_small
    out["mask_2d_small"] = np.stack([du.get_inpaint_mask(b.copy(), l2i_small, 90, 160, 0.1, use_3d_edit_mask=False).numpy() for b in fwd[:6]])
    out["bbox_2d_small"] = np.stack([du.get_2d_bbox(b.copy(), l2i_small, 90, 160, 0.1) for b in fwd[:6]])

    path = os.path.join(HERE, "data_side.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
