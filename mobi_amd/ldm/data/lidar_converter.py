"""Range-view <-> point-cloud transforms of the sampling harness (reference: ldm/data/lidar_converter.py --
pool_resize :8-19, LidarConverter.range2pcd :122-172, resize :230-290, undo_default_transforms :436-485).

Only the inference-side methods are here; the dataset-side ones (pcd2range, tile, bbox_crop, apply_default_transforms)
belong to the data loader."""
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
        if log_scale:
            raise NotImplementedError("MObI's configs use the linear depth scale")
        self.current_H, self.current_W = H, W
        self.base_size = (H, W)
        self.depth_interval = depth_interval
        self.log_scale, self.depth_scale = log_scale, depth_scale

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
