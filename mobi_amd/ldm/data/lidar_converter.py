"""Range-view <-> point-cloud transforms (reference: ldm/data/lidar_converter.py -- pool_resize :8-19,
LidarConverter.pcd2range :38-120, range2pcd :122-172, get_range_coords :174-228, resize :230-290, tile :292-324,
bbox_crop :326-385, apply_default_transforms :387-434, undo_default_transforms :436-485).

Sampling side (undo_default_transforms, range2pcd) on the device; dataset side (one 32 x 1096 sweep and eight box
corners per call, in DataLoader workers) in numpy with the reference's own operation order, so that indices and box
coordinates come out bit for bit.  The batched device form of the dataset-side pixel work is `ops.range_prepare`
(nuscenes.py: `NuScenesDataset.collate_device`)."""
import numpy as np
import torch
import torch.nn.functional as F

from ... import ops


def pool_resize(x, size, mode="avg_pool"):
    x = x.float()
    _, _, height, width = x.shape
    k = (height // size[0], width // size[1])
    if mode == "avg_pool":
        return F.avg_pool2d(x, kernel_size=k)
    if mode == "max_pool":
        return F.max_pool2d(x, kernel_size=k)
    raise NotImplementedError


class LidarConverter:
    def __init__(self, H=32, W=1096, depth_interval=(1.4, 54), log_scale=False, depth_scale=5.8):
        self.current_H, self.current_W = H, W
        self.base_size = (H, W)
        self.depth_interval = depth_interval
        self.log_scale, self.depth_scale = log_scale, depth_scale
        self.beam_pitch_angles = np.array([0.0232 * x for x in range(-23, 9)])        # nuScenes' 32 beams

    # ---- dataset side -------------------------------------------------------------------------------------------------
    def _beam_row(self, pitch):
        lo, hi = self.beam_pitch_angles.min(), self.beam_pitch_angles.max()
        row = (pitch - lo) / (hi - lo) * 31
        return 31 - np.round(np.clip(row, 0, 31)).astype(np.int32)

    def _encode_depth(self, metres):
        d = np.log2(metres + 0.0001 + 1) / self.depth_scale if self.log_scale else metres / self.depth_interval[1]
        return np.clip(d * 2.0 - 1.0, -1, 1)

    def pcd2range(self, pcd, label=None):
        """points [N, 3] (+ label [N]) -> (range_depth, range_int | None, kept [N] bool, range_pitch, range_yaw), each
        (H, W): the NEAREST point of a pixel wins (points are written far to near); empty pixels: depth code -1, the
        beam's nominal pitch, the column's nominal yaw."""
        pts = np.array(pcd)
        lab = None if label is None else np.array(label)
        dist = np.linalg.norm(pts, 2, axis=1)
        kept = np.logical_and(dist > self.depth_interval[0], dist < self.depth_interval[1])
        dist, pts = dist[kept], pts[kept]
        yaw = -np.arctan2(pts[:, 1], pts[:, 0])
        pitch = np.arcsin(pts[:, 2] / dist)
        rows = self._beam_row(pitch)
        cols = 0.5 * (yaw / np.pi + 1.0) * self.current_W
        cols = np.maximum(0, np.minimum(self.base_size[1] - 1, np.floor(cols))).astype(np.int32)
        far_to_near = np.argsort(dist)[::-1]
        rows, cols = rows[far_to_near], cols[far_to_near]
        dist, pitch, yaw = dist[far_to_near], pitch[far_to_near], yaw[far_to_near]

        H, W = self.base_size
        col_frac = np.meshgrid(np.arange(W), np.arange(H))[0].astype(np.float32) / W
        range_yaw = np.pi * (col_frac * 2 - 1)
        range_pitch = np.zeros((H, W), dtype=np.float32)
        range_pitch[:] = self.beam_pitch_angles[::-1][:H, None] if H == 32 else 0
        range_depth = np.full((H, W), -1, dtype=np.float32)
        range_depth[rows, cols] = dist
        range_pitch[rows, cols] = pitch
        range_yaw[rows, cols] = yaw
        range_int = None
        if lab is not None:
            range_int = np.full((H, W), 0, dtype=np.float32)
            range_int[rows, cols] = lab[kept][far_to_near]
        range_depth = np.where(range_depth < 0, 0, range_depth)
        return self._encode_depth(range_depth), range_int, kept, range_pitch, range_yaw

    def get_range_coords(self, bbox_3d):
        """[8, 3] box corners -> [8, 3] (column, row, depth code) in the CURRENT view width; columns are measured from
        the box centre's azimuth, so a box across the +-pi seam stays contiguous."""
        box = np.array(bbox_3d)
        dist = np.linalg.norm(box, 2, axis=1)
        yaw_c = -np.arctan2(np.mean(box[:, 1]), np.mean(box[:, 0]))
        c, s = np.cos(yaw_c), np.sin(yaw_c)
        turned = np.dot(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]), box.T).T
        yaw = -(np.arctan2(turned[:, 1], turned[:, 0]) - yaw_c)
        pitch = np.arcsin(turned[:, 2] / dist)
        cols = 0.5 * (yaw / np.pi + 1.0)
        rows = self._beam_row(pitch)
        cols *= self.current_W
        return np.concatenate([cols[:, None], rows[:, None], self._encode_depth(dist)[:, None]], axis=-1)

    @staticmethod
    def _nearest(a, new_H, new_W):
        """cv2.resize(a, (new_W, new_H), interpolation=cv2.INTER_NEAREST) restated (OpenCV resizeNN: source index =
        min(floor(dst * (1 / (dst_size / src_size))), src_size - 1), in double)."""
        h, w = a.shape
        ys = np.minimum(np.floor(np.arange(new_H) * (1.0 / (new_H / h))).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(new_W) * (1.0 / (new_W / w))).astype(np.int64), w - 1)
        return a[ys[:, None], xs[None, :]]

    def _resized(self, a, new_H, new_W, pool):
        if a is None or a.shape == (new_H, new_W):
            return a
        if a.shape[0] % new_H == 0 and a.shape[1] % new_W == 0:              # whole-factor reduction: pooling
            return pool_resize(torch.from_numpy(a)[None, None, ...], (new_H, new_W), mode=pool).squeeze().numpy()
        return self._nearest(a, new_H, new_W)

    def resize(self, range_depth=None, range_int=None, mask=None, bbox_range_coords=None, new_W=1096, new_H=32):
        cp = lambda a: None if a is None else a.copy()
        range_depth = self._resized(cp(range_depth), new_H, new_W, "avg_pool")
        range_int = self._resized(cp(range_int), new_H, new_W, "avg_pool")
        mask = self._resized(cp(mask), new_H, new_W, "max_pool")
        coords = cp(bbox_range_coords)
        if coords is not None:
            coords[:, 0] = coords[:, 0] * new_W / self.current_W
            coords[:, 1] = coords[:, 1] * new_H / self.current_H
        self.current_W, self.current_H = new_W, new_H
        return range_depth, range_int, mask, coords

    def tile(self, range_depth=None, range_int=None, mask=None, bbox_range_coords=None, n=3):
        """n sweeps side by side (a window around any azimuth never meets an edge); box columns move to the middle
        copy IN PLACE, as the reference's does."""
        rep = lambda a: None if a is None else np.tile(a, n)
        if bbox_range_coords is not None:
            bbox_range_coords[:, 0] += self.current_W
        self.current_W *= n
        return rep(range_depth), rep(range_int), rep(mask), bbox_range_coords

    def bbox_crop(self, bbox_range_coords, range_depth=None, range_int=None, mask=None, width=512, random_crop=False,
                  crop_left=None):
        import random
        assert bbox_range_coords is not None
        centre = int(np.mean(bbox_range_coords[:, 0]))
        if crop_left is not None:
            before = centre - crop_left
        elif random_crop:
            before = random.randint(width // 4, width - width // 4)
        else:
            before = width // 2
        x0, x1 = centre - before, centre + (width - before)
        cut = lambda a: None if a is None else a.copy()[:, x0:x1]
        coords = bbox_range_coords.copy() - np.array([x0, 0, 0])
        self.current_W = width
        return cut(range_depth), cut(range_int), cut(mask), coords, x0

    def apply_default_transforms(self, bbox_range_coords, range_depth=None, range_int=None, mask=None, height=512,
                                 width=512, crop_left=None, width_crop=None, random_crop=False):
        """The dataset's view of one object: 3 sweeps side by side -> a window of `width_crop` columns around the box (a
        power of two >= 1.5 box widths, 64 ... width) -> height x width.  Returns (depth, intensity, mask, coords,
        crop_left, width_crop)."""
        range_depth, range_int, mask, coords = self.tile(range_depth, range_int, mask, bbox_range_coords, n=3)
        if width_crop is None:
            extent = coords[:, 0].max() - coords[:, 0].min()
            width_crop = max(64, min(width, int(2 ** np.ceil(np.log2(extent * 1.5)))))
        range_depth, range_int, mask, coords, crop_left = self.bbox_crop(
            coords, range_depth, range_int, mask, width=width_crop, crop_left=crop_left, random_crop=random_crop)
        range_depth, range_int, mask, coords = self.resize(range_depth, range_int, mask, coords, new_W=width, new_H=height)
        return range_depth, range_int, mask, coords, crop_left, width_crop

    # ---- sampling side ------------------------------------------------------------------------------------------------

    def undo_default_transforms(self, crop_left, width_crop, range_depth_crop, range_depth, range_int_crop=None,
                                range_int=None, mask=None):
        """One sample: (H_c, W_c) crop back into the (H0, W0) sweep.  Arrays may be numpy (result numpy, as the
        reference) or device tensors (result tensors)."""
        if mask is not None:
            raise NotImplementedError("masked un-crop is a dataset-side path")
        assert range_int is None or range_int_crop is not None
        as_numpy = not isinstance(range_depth_crop, torch.Tensor)
        dev = torch.device("cuda") if as_numpy else range_depth_crop.device
        t = lambda a: None if a is None else torch.as_tensor(np.asarray(a) if as_numpy else a).to(dev, torch.float32)[None]
        out = ops.range_paste(t(range_depth_crop), t(range_depth), [int(crop_left)], [int(width_crop)],
                              sample_int=t(range_int_crop), int_orig=t(range_int))
        d, i = out["depth_unc"][0], (out["int_unc"][0] if range_int is not None else None)
        if as_numpy:
            return d.cpu().numpy(), (None if i is None else i.cpu().numpy())
        return d, i

    def range2pcd(self, range_depth, range_pitch, range_yaw, label=None):
        """(H, W) range view -> (points [N, 3] fp32, label [N] | None, beam_index [N]); numpy in, numpy out."""
        depth = np.asarray(range_depth, dtype=np.float32)
        if depth.shape != tuple(self.base_size):
            raise NotImplementedError("range2pcd is used at the sweep's own resolution on the sampling path")
        depth = (depth + 1) / 2
        depth = depth * self.depth_interval[1]
        d = depth.flatten()
        yaw, pitch = np.asarray(range_yaw).flatten(), np.asarray(range_pitch).flatten()
        pcd = np.zeros((len(yaw), 3)).astype(np.float32)
        pcd[:, 0] = np.cos(yaw) * np.cos(pitch) * d
        pcd[:, 1] = -np.sin(yaw) * np.cos(pitch) * d
        pcd[:, 2] = np.sin(pitch) * d
        valid = np.logical_and(d > self.depth_interval[0], d < self.depth_interval[1])
        pcd = pcd[valid, :]
        lab = np.asarray(label).flatten()[valid] if label is not None else None
        h, w = np.asarray(range_pitch).shape
        beam = np.tile(np.arange(h - 1, -1, -1).reshape(h, 1), (1, w)).flatten()[valid]
        return pcd, lab, beam
