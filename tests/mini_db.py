"""A miniature object / scene database in the reference's on-disk formats (bevfusion/tools/data_converter/
create_pbe_database.py:116-139, 234-253): `dbinfos.csv`, `scene_infos.pkl`, per-scene `sample-*_range_{depth,intensity,
pitch,yaw,instance_mask}.npy` and camera JPEGs.  Synthetic content: a ground plane, a few boxes with lidar returns on
their faces, noise images.  Test infrastructure only."""
import os
import pickle

import numpy as np
import pandas as pd

CORNER_SIGNS = np.array([[-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1]], dtype=np.float64)


def _box(centre, half, heading):
    rot = np.array([[np.cos(heading), -np.sin(heading), 0], [np.sin(heading), np.cos(heading), 0], [0, 0, 1]])
    return (CORNER_SIGNS * half) @ rot.T + centre


def _sweep(rng, boxes):
    """Points of a 32-beam sweep: every beam / azimuth hits the ground or a box face (coarse ray casting)."""
    az = np.repeat(np.linspace(-np.pi, np.pi, 1096, endpoint=False) + 1e-3, 32)
    pitch = np.tile(0.0232 * (np.arange(32) - 23), 1096) + rng.normal(0, 5e-4, 32 * 1096)
    dirs = np.stack([np.cos(pitch) * np.cos(az), np.cos(pitch) * np.sin(az), np.sin(pitch)], 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(dirs[:, 2] < -1e-3, -1.8 / dirs[:, 2], 80.0)                  # ground 1.8 m below the sensor
    owner = np.full(len(t), -1)
    for k, b in enumerate(boxes):                                                    # boxes as bounding spheres (coarse)
        c, r = b.mean(0), 0.5 * np.linalg.norm(b.max(0) - b.min(0)) * 0.6
        proj = dirs @ c
        d2 = c @ c - proj ** 2
        hit = (d2 < r * r) & (proj > 0)
        th = proj - np.sqrt(np.maximum(r * r - d2, 0))
        closer = hit & (th < t)
        t[closer], owner[closer] = th[closer], k
    pts = (dirs * t[:, None]).astype(np.float32)
    inten = rng.integers(0, 256, len(t)).astype(np.float32)
    return pts, inten, owner


def build(root, n_scenes=2, seed=0, image_hw=(450, 800)):
    from PIL import Image
    from mobi_amd.ldm.data.lidar_converter import LidarConverter
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    H, W = image_hw
    K = np.array([[0.79 * W, 0, W / 2, 0], [0, 0.79 * W, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    l2c = np.array([[0, -1, 0, 0.0], [0, 0, -1, -0.2], [1, 0, 0, -0.3], [0, 0, 0, 1]], dtype=np.float64)
    l2i = K @ l2c
    scenes, rows = {}, []
    for s in range(n_scenes):
        token = f"scene{s:02d}"
        classes = ["car", "pedestrian", "car"]
        boxes = np.stack([_box(np.array([8.0 + 3 * s, -2.5, -0.8]), np.array([2.2, 0.9, 0.8]), 0.3),
                          _box(np.array([6.0, 1.8 + s, -0.9]), np.array([0.35, 0.35, 0.9]), 0.0),
                          _box(np.array([14.0, 4.0, -0.7]), np.array([2.0, 0.9, 0.8]), -0.5 + s)])
        pts, inten, owner = _sweep(rng, boxes)
        conv = LidarConverter()
        depth, rint, kept, pitch, yaw = conv.pcd2range(pts, inten)
        _, inst, _, _, _ = conv.pcd2range(pts, owner.astype(np.float32) + 1)        # 0 = nothing, k + 1 = box k
        paths = {}
        for name, arr in (("depth", depth), ("intensity", rint), ("pitch", pitch), ("yaw", yaw), ("instance_mask", inst - 1)):
            paths[name] = os.path.join(root, f"sample-{token}_range_{name}.npy")
            np.save(paths[name], arr.astype(np.float32))
        img_path = os.path.join(root, f"{token}_CAM_FRONT.jpg")
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 255 // W), (yy * 255 // H), ((xx + yy) % 256)], -1).astype(np.uint8)
        img = np.clip(img.astype(np.int32) + rng.integers(-20, 20, img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(img_path, quality=95)
        scenes[token] = {"sample_idx": token, "timestamp": 1538984834447585 + s, "gt_bboxes_3d_corners": boxes,
                         "range_depth_path": paths["depth"], "range_intensity_path": paths["intensity"],
                         "range_pitch_path": paths["pitch"], "range_yaw_path": paths["yaw"],
                         "range_instance_mask_path": paths["instance_mask"], "lidar2image_transforms": [l2i],
                         "lidar2camera_transforms": [l2c], "cam_types": ["CAM_FRONT"], "image_paths": [img_path],
                         "lidar_path": os.path.join(root, f"{token}_LIDAR_TOP.pcd.bin")}
        for k, cls in enumerate(classes):
            dist = np.sqrt(boxes[k, :, 0] ** 2 + boxes[k, :, 1] ** 2)
            rows.append({"track_id": f"track{s}{k}", "scene_token": token, "timestamp": 1538984834447585 + s, "cam_type": "CAM_FRONT",
                         "cam_idx": 0, "scene_obj_idx": k, "object_class": cls, "camera_visibility_mask": 0.9,
                         "max_iou_overlap": 0.1, "reference_image_h": 150, "reference_image_w": 200,
                         "num_lidar_points": int((owner == k).sum()), "city": "boston", "is_raining": False, "is_night": bool(s % 2),
                         "is_erase_box": False, "max_distance": dist.max(), "min_distance": dist.min()})
    csv, pkl = os.path.join(root, "dbinfos.csv"), os.path.join(root, "scene_infos.pkl")
    pd.DataFrame(rows).to_csv(csv)
    with open(pkl, "wb") as f:
        pickle.dump(scenes, f)
    return csv, pkl
