"""`ldm.data.nuscenes.NuScenesDataset` for the engine: one edited object per item, batch schema A0 of SURVEY.md section 8
(reference: ldm/data/nuscenes.py -- get_tensor / get_tensor_clip :30-49, NuScenesDataset.__init__ :52-252, __getitem__
:254-310, get_reference :317-379, get_id_name :381-394, get_range_data :396-493, get_image_data :495-594).

It reads the reference's own on-disk formats (bevfusion/tools/data_converter/create_pbe_database.py:116-139, 234-253):
`*_dbinfos_pbe_*.csv` (one row per object: track, scene token, camera index, class, visibility, box statistics),
`*_scene_infos_pbe_*.pkl` (per scene: box corners, lidar2image / lidar2camera, image paths, the `sample-*_range_{depth,
intensity,pitch,yaw,instance_mask}.npy` paths) and the nuScenes camera JPEGs.

Host side (what a DataLoader worker runs, numpy / torch-CPU like the reference): file IO, the 8-corner geometry, the
crop windows.  Per-pixel work has a batched device form, `NuScenesDataset.collate_device`: raw sweeps and masks go to
the GPU once per batch and `ops.range_prepare` builds `range_data` / `range_data_inpaint` / `range_instance_mask` for the
whole batch in one launch.

Third-party pieces of the reference that are absent here are restated from their documentation (parity unpinned for
exactly these): torchvision 0.11 `ToTensor` / `Normalize` / tensor `Resize` (bilinear, align_corners=False, no
antialias), albumentations 0.4 `Resize` = `cv2.resize(..., INTER_LINEAR)` on uint8 (`resize_linear_u8`), cv2.fillPoly
(`utils.fill_box_faces`), cv2.resize INTER_NEAREST (`LidarConverter._nearest`).  The reference-image augmentations
(`ref_aug=True`: flip / rotate / blur / brightness-contrast) belong to the training row and are not built."""
import json
import os
import pickle
import random
import warnings

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F
import torch.utils.data as data

from .lidar_converter import LidarConverter
from .utils import (depth_normalization, expand_bbox_corners, fill_box_faces, get_2d_bbox, get_image_coords,
                    get_inpaint_mask, rotate_bbox, translate_bbox)

CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def _to_tensor(pic):
    """torchvision.transforms.ToTensor: uint8 HWC -> float CHW / 255; other dtypes keep their values; HW gets a channel."""
    a = np.asarray(pic)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.float().div(255) if a.dtype == np.uint8 else t


class _Pipeline:
    def __init__(self, to_tensor, mean=None, std=None):
        self.to_tensor, self.mean, self.std = to_tensor, mean, std

    def __call__(self, x):
        if self.to_tensor:
            x = _to_tensor(x)
        if self.mean is not None:
            m = torch.as_tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            x = (x - m) / s
        return x


def get_tensor(normalize=True, toTensor=True):
    return _Pipeline(toTensor, (0.5, 0.5, 0.5) if normalize else None, (0.5, 0.5, 0.5) if normalize else None)


def get_tensor_clip(normalize=True, toTensor=True):
    return _Pipeline(toTensor, CLIP_MEAN if normalize else None, CLIP_STD if normalize else None)


def resize_linear_u8(img, height, width):
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR) for uint8 HWC, restated from OpenCV's resize.cpp:
    sample position (d + 0.5) * scale - 0.5, neighbours clamped to the image, weights rounded to 11-bit integers, the
    vertical pass `(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2` on rows held at 2^11 scale."""
    src = np.asarray(img)
    h, w = src.shape[:2]

    def taps(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        low = i0 < 0
        i0[low], frac[low] = 0, 0.0
        high = i0 >= n_src - 1
        i0[high], frac[high] = n_src - 1, 0.0
        i1 = np.minimum(i0 + 1, n_src - 1)
        w1 = np.rint(frac * 2048.0).astype(np.int64)
        w0 = np.rint((1.0 - frac) * 2048.0).astype(np.int64)
        return i0, i1, w0, w1
    x0, x1, a0, a1 = taps(width, w)
    y0, y1, b0, b1 = taps(height, h)
    s = src.astype(np.int64)
    rows = s[:, x0] * a0[None, :, None] + s[:, x1] * a1[None, :, None]                     # [h, width, C] at 2^11
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class NuScenesDataset(data.Dataset):
    def __init__(self, state, object_database_path, scene_database_path, object_classes, expand_mask_ratio=0,
                 expand_ref_ratio=0, ref_aug=True, prob_use_3d_edit_mask=1, prob_drop_context=0, ref_mode="id-ref",
                 image_height=512, image_width=512, range_height=512, range_width=512, reference_image_min_h=100,
                 reference_image_max_h=800, reference_image_min_w=100, reference_image_max_w=1400, frustum_iou_max=0.5,
                 camera_visibility_min=0.7, object_area_crop=0.2, object_random_crop=True, min_lidar_points=64,
                 rot_every_angle=0, rot_test_scene=None, rot_test_cam_idx=3, rot_test_bbox_coord=[3, -10, -1.5],
                 use_lidar=False, use_camera=True, random_range_crop=False, num_samples_per_class=None, prob_erase_box=0,
                 fixed_sampling=True, sample_each_frame=False, return_original_image=False, range_object_norm=True,
                 range_object_norm_scale=0.75, range_int_norm=False, object_meta_dump_path=None, specific_object=None):
        if ref_aug:
            raise NotImplementedError("reference-image augmentation (ref_aug=True) is the training row's; the sampling "
                                      "configs (data.params.test / rotation_test) use ref_aug: false")
        self.state, self.ref_aug, self.ref_mode = state, ref_aug, ref_mode
        self.expand_mask_ratio, self.expand_ref_ratio = expand_mask_ratio, expand_ref_ratio
        self.prob_use_3d_edit_mask, self.prob_drop_context = prob_use_3d_edit_mask, prob_drop_context
        self.rot_test_scene, self.rot_test_cam_idx = rot_test_scene, rot_test_cam_idx
        self.rot_test_bbox_coord = np.array(rot_test_bbox_coord)
        self.use_lidar, self.use_camera = use_lidar, use_camera
        self.random_range_crop, self.object_area_crop = random_range_crop, object_area_crop
        self.object_random_crop, self.return_original_image = object_random_crop, return_original_image
        self.range_object_norm, self.range_object_norm_scale = range_object_norm, range_object_norm_scale
        self.range_int_norm = range_int_norm
        self.num_samples_per_class, self.prob_erase_box, self.fixed_sampling = num_samples_per_class, prob_erase_box, fixed_sampling
        self.image_height, self.image_width = image_height, image_width
        self.range_height, self.range_width = range_height, range_width
        self.object_classes = list(object_classes)

        db = pd.read_csv(object_database_path, index_col=0)
        db = db[db["object_class"].isin(self.object_classes) & (db["camera_visibility_mask"] >= camera_visibility_min) &
                (db["max_distance"] < 54) & (db["min_distance"] > 1.4)]
        self.objects_meta_orig = db
        usable = db[(db["reference_image_h"] >= reference_image_min_h) & (db["reference_image_h"] <= reference_image_max_h) &
                    (db["reference_image_w"] >= reference_image_min_w) & (db["reference_image_w"] <= reference_image_max_w) &
                    (db["max_iou_overlap"] <= frustum_iou_max) & (db["num_lidar_points"] >= min_lidar_points)]
        self.erase_meta_all = usable[usable["is_erase_box"]]
        self.objects_meta_all = usable[~usable["is_erase_box"]]

        warnings.filterwarnings("ignore", category=FutureWarning, message=".*DataFrameGroupBy.apply operated on the grouping columns.*")
        if specific_object is not None:
            # "sample-<scene>_track-<track>_time-<timestamp>_<class>_<ref mode>_rot-<angle>..."
            parts = specific_object.split("_")
            scene, track, stamp = parts[0].split("-")[1], parts[1].split("-")[1], int(parts[2].split("-")[1])
            self.objects_meta = db[(db["track_id"] == track) & (db["scene_token"] == scene) & (db["timestamp"] == stamp)]
            self.num_samples_per_class = None
        else:
            if sample_each_frame:
                picked = self.objects_meta_all.groupby("scene_token").apply(lambda g: g.sample(n=1))
                missing = set(db["scene_token"]) - set(picked["scene_token"])
                extra = (db[db["scene_token"].isin(missing) & ~db["is_erase_box"]].groupby("scene_token")
                         .apply(lambda g: g.nlargest(3, "num_lidar_points").sample(n=1)).reset_index(drop=True))
                self.objects_meta = pd.concat([picked, extra])
                self.objects_meta_all = pd.concat([self.objects_meta_all, extra])
                self.num_samples_per_class = None
            elif num_samples_per_class is not None and fixed_sampling:
                n = num_samples_per_class
                self.objects_meta = self.objects_meta_all.groupby("object_class").apply(
                    lambda g: g.sample(n, replace=(len(g) < n)))
            else:
                self.objects_meta = self.objects_meta_all
            self.objects_meta = self.objects_meta.reset_index(drop=True)

        if object_meta_dump_path is not None:
            os.makedirs(os.path.dirname(object_meta_dump_path), exist_ok=True)
            with open(object_meta_dump_path, "w") as f:
                json.dump({row["scene_token"]: row["track_id"] for _, row in self.objects_meta.iterrows()}, f)

        by_class = lambda frame: [frame[frame["object_class"] == c].index.tolist() for c in self.object_classes]
        self.idx_lists, self.idx_lists_erase = by_class(self.objects_meta), by_class(self.erase_meta_all)

        if rot_every_angle != 0:
            angles = np.arange(0, 360, rot_every_angle)
            self.objects_meta = pd.concat([self.objects_meta] * len(angles), ignore_index=True)
            self.objects_meta["bbox_rot_angle"] = np.repeat(angles, len(self.objects_meta) // len(angles))
            if self.num_samples_per_class is not None:
                self.num_samples_per_class *= len(angles)

        with open(scene_database_path, "rb") as f:
            self.scenes_info = pickle.load(f)

    # ------------------------------------------------------------------------------------------------------------------
    def __len__(self):
        if self.num_samples_per_class is None:
            return len(self.objects_meta)
        return len(self.object_classes) * self.num_samples_per_class

    def _pick(self, index):
        slot = index % len(self.object_classes)
        if random.random() < self.prob_erase_box and len(self.idx_lists_erase[slot]) > 0:
            index = np.random.choice(self.idx_lists_erase[slot])
            return index, self.erase_meta_all.loc[index]
        if self.num_samples_per_class and self.fixed_sampling is False:
            index = np.random.choice(self.idx_lists[slot])
        return index, self.objects_meta.loc[index]

    def __getitem__(self, index):
        return self._build(index, raw=False)

    def raw_item(self, index):
        """`__getitem__` without the range view's per-pixel work: the lidar part carries the untouched sweep, the crop
        window and the edit region's corner pixels; `collate_device` finishes a list of these on the GPU."""
        return self._build(index, raw=True)

    def _build(self, index, raw):
        index, meta = self._pick(index)
        if self.rot_test_scene is not None:
            scene, cam_idx = self.scenes_info[self.rot_test_scene], self.rot_test_cam_idx
        else:
            scene, cam_idx = self.scenes_info[meta["scene_token"]], meta["cam_idx"]
        ref_image, ref_box, ref_class = self.get_reference(meta, index)
        if self.rot_test_scene is None:
            box = scene["gt_bboxes_3d_corners"][meta["scene_obj_idx"]]
        else:
            box = translate_bbox(ref_box, self.rot_test_bbox_coord)
        box = rotate_bbox(box, meta.get("bbox_rot_angle", 0))
        item = {"id_name": self.get_id_name(meta), "bbox_3d": box, "ref_class": ref_class, "image": {}, "lidar": {}}
        if self.use_camera:
            item["image"] = self.get_image_data(scene, cam_idx, box, raw=raw)
            item["image"]["cond"]["ref_image"] = ref_image
        if self.use_lidar:
            item["lidar"] = self.get_range_data(scene, box, meta["scene_obj_idx"], raw=raw)
            item["lidar"]["cond"]["ref_image"] = ref_image
            if self.use_camera and not raw:                  # the camera box token carries the RANGE-view depth code
                item["image"]["cond"]["ref_bbox"][..., 2] = item["lidar"]["cond"]["ref_bbox"][..., 2]
        erase = bool(meta["is_erase_box"] or self.ref_mode == "erase-ref")
        if raw:
            item["erase"] = erase                            # (the camera token is finished by `collate_device`)
        if erase:
            if self.use_lidar and not raw:
                item["image"]["cond"]["ref_bbox"] *= 0
            if self.use_camera:
                item["lidar"]["cond"]["ref_bbox"] *= 0
        return item

    # ------------------------------------------------------------------------------------------------------------------
    def _reference_meta(self, meta, index):
        pool = self.objects_meta_all
        if self.ref_mode in ("id-ref", "erase-ref") or meta["is_erase_box"]:
            return meta
        same_class = pool["object_class"] == meta["object_class"]
        same_domain = (pool["is_raining"] == meta["is_raining"]) & (pool["is_night"] == meta["is_night"])
        if self.ref_mode == "in-domain-ref":
            return pool[same_class & same_domain].sample(1, random_state=index).iloc[0]
        if self.ref_mode == "cross-domain-ref":
            return pool[same_class & ~same_domain].sample(1, random_state=index).iloc[0]
        if self.ref_mode == "track-ref":
            tracked = pool[pool["track_id"] == meta["track_id"]]
            if len(tracked):
                return tracked.iloc[0]
            from scipy.stats import beta                 # (unreachable with an empty selection, as in the reference)
            gaps = abs(tracked.timestamp - meta.timestamp)
            gaps /= max(gaps)
            w = beta.pdf(gaps, 4, 1)
            return tracked.iloc[np.random.choice(len(gaps), p=w / np.sum(w))]
        raise ValueError("Invalid ref_mode")

    def _frame(self, path):
        """RGB uint8 [H, W, 3] of a camera JPEG; the last frame stays decoded (with `id-ref` the reference patch and the
        edited view come from the same file: one decode instead of two)."""
        from PIL import Image
        hit = self.__dict__.get("_last_frame")
        if hit is None or hit[0] != path:
            hit = (path, np.array(Image.open(path).convert("RGB")))
            self.__dict__["_last_frame"] = hit
        return hit[1]

    def get_reference(self, current_object_meta, index):
        meta = self._reference_meta(current_object_meta, index)
        scene = self.scenes_info[meta["scene_token"]]
        cam = meta["cam_idx"]
        box = scene["gt_bboxes_3d_corners"][meta["scene_obj_idx"]]
        ref_class = meta["object_class"]
        if self.ref_mode == "erase-ref" or current_object_meta["is_erase_box"]:
            patch, ref_class = np.zeros((224, 224, 3), dtype=np.uint8), "empty"
        else:
            frame = self._frame(scene["image_paths"][cam])
            H, W = frame.shape[:2]
            x1, y1, x2, y2 = get_2d_bbox(box, scene["lidar2image_transforms"][cam], H, W, self.expand_ref_ratio)
            w, h = np.maximum(x2 - x1 + 1, 1), np.maximum(y2 - y1 + 1, 1)
            patch = frame[y1:y1 + h, x1:x1 + w]
        patch = resize_linear_u8(patch, 224, 224)
        return get_tensor_clip()(patch), box, ref_class

    def get_id_name(self, object_meta):
        name = "sample-{}_track-{}_time-{}_{}_{}_rot-{}".format(
            object_meta["scene_token"], object_meta["track_id"], object_meta["timestamp"], object_meta["object_class"],
            self.ref_mode, object_meta.get("bbox_rot_angle", 0))
        return name + "-aug" if self.ref_aug else name

    # ------------------------------------------------------------------------------------------------------------------
    def _load_sweep(self, scene_info, obj_idx):
        if "range_depth_path" in scene_info and "range_intensity_path" in scene_info:
            depth, inten = np.load(scene_info["range_depth_path"]), np.load(scene_info["range_intensity_path"])
            pitch, yaw = np.load(scene_info["range_pitch_path"]), np.load(scene_info["range_yaw_path"])
            if "range_instance_mask_path" in scene_info:
                inst = (np.load(scene_info["range_instance_mask_path"]) == obj_idx).astype(np.float32)
            else:
                inst = np.zeros_like(depth).astype(np.float32)
                warnings.warn("No instance mask found")
            return depth, inten, pitch, yaw, inst
        if "lidar_path" in scene_info:
            scan = np.load(scene_info["lidar_path"])
            depth, inten, _, pitch, yaw = LidarConverter().pcd2range(scan[:, :3].astype(np.float32), label=scan[:, 3])
            return depth, inten, pitch, yaw, np.zeros_like(depth).astype(np.float32)
        raise ValueError("No lidar data found")

    def _box_token_and_depth_range(self, coords):
        tok = torch.tensor(coords).float()
        tok[..., 0] /= self.range_width
        tok[..., 1] /= self.range_height
        span = tok[:, 2].max() - tok[:, 2].min()
        lo = torch.clamp(tok[:, 2].min() - 0.1 * span, -1, 1)
        hi = torch.clamp(tok[:, 2].max() + 0.1 * span, -1, 1)
        return tok, lo, hi

    def get_range_data(self, scene_info, bbox_3d, obj_idx, raw=False):
        """raw=True: the sweep and the crop window only (for `collate_device`); else the finished tensors."""
        depth0, int0, pitch, yaw, inst0 = self._load_sweep(scene_info, obj_idx)
        conv = LidarConverter()
        coords = conv.get_range_coords(bbox_3d)
        if raw:
            _, _, _, coords, shift_left, width_crop = conv.apply_default_transforms(
                coords, height=self.range_height, width=self.range_width, random_crop=self.random_range_crop)
            depth = inten = inst = None
        else:
            depth, inten, inst, coords, shift_left, width_crop = conv.apply_default_transforms(
                range_depth=depth0.copy(), range_int=int0.copy(), mask=inst0.copy(), bbox_range_coords=coords,
                height=self.range_height, width=self.range_width, random_crop=self.random_range_crop)
        tok, lo, hi = self._box_token_and_depth_range(coords)
        if self.range_object_norm:
            tok[..., 2] = depth_normalization(tok[..., 2], lo, hi, alpha=self.range_object_norm_scale)
        # edit region: the faces of the expanded box in the same view (data/utils.py:174-198)
        mconv = LidarConverter()
        _, _, _, mcoords, _, _ = mconv.apply_default_transforms(
            mconv.get_range_coords(expand_bbox_corners(bbox_3d, self.expand_mask_ratio)), height=self.range_height,
            width=self.range_width, crop_left=shift_left, width_crop=width_crop)
        out = {"range_depth_orig": depth0.copy(), "range_int_orig": int0.copy(), "range_instance_mask_orig": inst0.copy(),
               "range_shift_left": shift_left, "width_crop": width_crop, "range_pitch": pitch,
               "range_yaw": yaw, "min_depth_obj": lo, "max_depth_obj": hi, "cond": {"ref_bbox": tok},
               "file_name": scene_info["lidar_path"].split("/")[-1]}
        if raw:
            out["range_mask_corners"] = torch.from_numpy(mcoords[:, :2].astype(np.int32))
            return out
        mask = (1. - torch.tensor(fill_box_faces(mcoords[:, :2].astype(np.int32), self.range_height, self.range_width) > 0.5).float()).unsqueeze(0)
        out["range_mask"] = mask
        depth = _to_tensor(depth)
        if self.range_object_norm:
            depth = depth_normalization(depth, lo, hi, alpha=self.range_object_norm_scale)
        inten = _to_tensor(((inten / 255) - 0.5) * 2)
        if self.range_int_norm:
            inten = 1 - torch.exp(-2 * (inten + 1))
            inten = torch.clamp(2 * inten - 1, -1, 1)
        range_data = torch.concat([depth, inten], dim=0)
        inpaint = range_data.clone() * mask
        if random.random() < self.prob_drop_context:
            inpaint *= 0
            range_data = range_data * (1 - mask)
        out.update(range_data=range_data, range_data_inpaint=inpaint, range_instance_mask=torch.tensor(inst).float().unsqueeze(0))
        return out

    # ------------------------------------------------------------------------------------------------------------------
    def _crop_window(self, count, x1, x2, y1, y2, H, W):
        """(left, top, crop_W, crop_H): a square in which the edit region (count pixels, extent x1..x2 / y1..y2) takes
        `object_area_crop` of the area, stretched where the region is longer, centred on it (or placed at random)."""
        side = int(np.sqrt(count / self.object_area_crop))
        crop_H = crop_W = side
        if y2 - y1 > crop_H:
            crop_W += crop_H - (y2 - y1)
            crop_H = y2 - y1
        if x2 - x1 > crop_W:
            crop_H += crop_W - (x2 - x1)
            crop_W = x2 - x1
        crop_H, crop_W = min(crop_H, H), min(crop_W, W)
        lo_x, hi_x = max(0, x2 - crop_W), min(x1, W - crop_W)
        lo_y, hi_y = max(0, y2 - crop_H), min(y1, H - crop_H)
        centred = ((lo_x + hi_x) // 2, (lo_y + hi_y) // 2)
        if self.object_random_crop:
            try:
                left, top = random.randint(lo_x, hi_x), random.randint(lo_y, hi_y)
            except Exception:
                left, top = centred
        else:
            left, top = centred
        return left, top, crop_W, crop_H

    def _edit_corners(self, bbox_3d, lidar2image, H, W):
        """int32 [8, 2]: the projected corners whose six faces are the edit region; with the 2D-box mask a flat box
        covering rows y1 .. y2 - 1, columns x1 .. x2 - 1 (off-image when empty)."""
        if random.random() < self.prob_use_3d_edit_mask:
            xy = get_image_coords(expand_bbox_corners(bbox_3d, self.expand_mask_ratio), lidar2image)
            return xy.astype(np.int32)
        x1, y1, x2, y2 = get_2d_bbox(bbox_3d, lidar2image, H, W, self.expand_mask_ratio)
        if x2 <= x1 or y2 <= y1:
            return np.full((8, 2), -16, dtype=np.int32)
        return np.array([[x1, y1], [x2 - 1, y1], [x2 - 1, y2 - 1], [x1, y2 - 1]] * 2, dtype=np.int32)

    def get_image_data(self, scene_info, cam_idx, bbox_3d, raw=False):
        lidar2image = scene_info["lidar2image_transforms"][cam_idx]
        path = scene_info["image_paths"][cam_idx]
        frame = self._frame(path)
        H, W = frame.shape[:2]
        token = get_image_coords(bbox_3d, lidar2image, include_depth=True)
        corners = self._edit_corners(bbox_3d, lidar2image, H, W)
        extra = {"file_name": path.split("/")[-1], "cam_type": scene_info["cam_types"][cam_idx], "lidar2image": lidar2image}
        if raw:                                               # the pixels are `collate_device`'s
            return {"frame": torch.from_numpy(frame), "mask_corners": torch.from_numpy(corners),
                    "token": torch.from_numpy(token), "cond": {}, "orig": extra}
        image = get_tensor()(frame)
        mask = 1. - torch.tensor(fill_box_faces(corners, H, W) > 0.5).float()
        if self.return_original_image:
            image_orig, mask_orig = image.clone(), mask.clone()
        if (mask == 1).all():
            mask = 1 - mask                                   # an erase box that projects nowhere: edit the whole frame
        hole = torch.nonzero(1 - mask)
        (y1, x1), (y2, x2) = (int(v) for v in hole.min(dim=0)[0]), (int(v) for v in hole.max(dim=0)[0])
        left, top, crop_W, crop_H = self._crop_window((1 - mask).sum().item(), x1, x2, y1, y2, H, W)

        image = image[:, top:top + crop_H, left:left + crop_W]
        mask = mask[top:top + crop_H, left:left + crop_W]
        token -= np.array([left, top, 0])
        token[..., 0] /= image.size(2)
        token[..., 1] /= image.size(1)
        token = torch.tensor(token).float()
        size = (self.image_height, self.image_width)
        image = F.interpolate(image[None], size=size, mode="bilinear", align_corners=False)[0]
        mask = F.interpolate(mask[None, None], size=size, mode="bilinear", align_corners=False)[0]
        inpaint = image.clone() * mask
        if random.random() < self.prob_drop_context:
            inpaint *= 0
            image = image * (1 - mask)
        out = {"GT": image, "inpaint_image": inpaint, "inpaint_mask": mask, "cond": {"ref_bbox": token}}
        if self.return_original_image:
            out["orig"] = dict(extra, crop=torch.tensor([left, top, crop_W, crop_H]), image=image_orig, mask=mask_orig)
        return out

    # ------------------------------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------------------------------
    def collate_device(self, items, device="cuda"):
        """Default-collate a list of `raw_item`s and finish the batch on the device.  Camera: `mobi_box_mask` gives every
        edit region's pixel count and extent (one small read-back), the host places the crop windows, `mobi_image_prepare`
        normalises / masks / crops / resizes all frames in one launch.  Lidar: `mobi_box_mask` rasterises the edit regions,
        `mobi_range_prepare` does tile x 3 -> window -> nearest resize, depth / intensity normalisation and the edit-mask
        product in one launch."""
        from torch.utils.data import default_collate
        from ... import ops
        batch = default_collate(items)
        move = lambda d: {k: move(v) if isinstance(v, dict) else (v.to(device) if isinstance(v, torch.Tensor) else v)
                          for k, v in d.items()}
        batch = move(batch)
        if self.prob_drop_context:
            raise NotImplementedError("context dropping is a training-side option")
        erase = batch.pop("erase")
        if self.use_camera:
            img = batch["image"]
            frames, corners, token = img.pop("frame"), img.pop("mask_corners"), img.pop("token").cpu().numpy()
            B, H, W, _ = frames.shape
            stats = ops.box_mask(corners, H, W, want_mask=False, want_stats=True).cpu().tolist()   # ONE read-back per batch
            crops, invert = [], []
            for count, x1, x2, y1, y2 in stats:
                invert.append(int(count == 0))                # (:513-515: nothing to edit -> the whole frame is the hole)
                if count == 0:
                    count, x1, x2, y1, y2 = H * W, 0, W - 1, 0, H - 1
                crops.append(self._crop_window(float(count), x1, x2, y1, y2, H, W))
            crop_t = torch.tensor(crops, dtype=torch.int32)
            img["GT"], img["inpaint_image"], img["inpaint_mask"] = ops.image_prepare(
                frames, corners, invert, crop_t, height=self.image_height, width=self.image_width)
            token = token - np.array([[[c[0], c[1], 0]] for c in crops], dtype=np.float64)
            token[..., 0] /= np.array([c[2] for c in crops], dtype=np.float64)[:, None]
            token[..., 1] /= np.array([c[3] for c in crops], dtype=np.float64)[:, None]
            tok = torch.tensor(token).float().to(device)
            if self.use_lidar:
                tok[..., 2] = batch["lidar"]["cond"]["ref_bbox"][..., 2]
                tok = tok * (1 - erase.to(device).float())[:, None, None]
            img["cond"]["ref_bbox"] = tok
            if self.return_original_image:
                full = ops.box_mask(corners, H, W)
                img["orig"].update(crop=crop_t.to(device).long(), mask=full,
                                   image=((frames.permute(0, 3, 1, 2).float().div(255) - 0.5) / 0.5))
            else:
                img.pop("orig")
        if not self.use_lidar:
            return batch
        lid = batch["lidar"]
        lid["range_mask"] = ops.box_mask(lid.pop("range_mask_corners"), self.range_height, self.range_width)[:, None]
        rd, rdi, inst = ops.range_prepare(
            lid["range_depth_orig"].float(), lid["range_int_orig"].float(), lid["range_instance_mask_orig"].float(),
            lid["range_shift_left"], lid["width_crop"], lid["min_depth_obj"].float(), lid["max_depth_obj"].float(),
            lid["range_mask"], height=self.range_height, width=self.range_width,
            alpha=self.range_object_norm_scale, object_norm=bool(self.range_object_norm), int_norm=bool(self.range_int_norm))
        lid.update(range_data=rd, range_data_inpaint=rdi, range_instance_mask=inst)
        return batch
