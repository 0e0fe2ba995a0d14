#!/usr/bin/env python3
"""tests/golden/data_item.npz: ONE WHOLE lidar item as the REFERENCE's `NuScenesDataset.get_range_data`
(ldm/data/nuscenes.py:396-493) builds it, on the miniature database of tests/mini_db.py (build container only).

The reference's dataset module imports here once the packages the image lacks are stubbed at import time (albumentations,
torchvision, cv2, PIL pieces: empty modules); its method is then called with a plain namespace as `self`.  Three library
calls inside it cannot run here and are replaced for the call; every field they touch is NOT stored (those stay "unpinned",
DESIGN section 9):
  * `cv2.resize(..., INTER_NEAREST)` of the 32-row crop to the output size (lidar_converter.py:263-283) -> a nearest
    resize restated from OpenCV's documentation: `range_data`, `range_data_inpaint`, `range_instance_mask` are not stored;
  * `get_tensor(normalize=False, toTensor=True)` (torchvision ToTensor) -> numpy -> tensor with a leading axis;
  * `get_range_inpaint_mask` (cv2.fillPoly) -> ones: `range_mask` is not stored.
Stored per (scene, object): every OTHER field of the item -- the untouched sweep (`range_*_orig`, pitch, yaw), the crop window
(`range_shift_left`, `width_crop`), the object's depth range and the 8-corner box token (`cond.ref_bbox`: tile / bbox_crop /
resize of the coordinates, the division by the view size, `depth_normalization`), all the reference's own arithmetic.
      python tests/golden/make_golden_data_item.py"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from tests import mini_db                                  # noqa: E402  (builds the files with the repo's converter: input data)
import make_golden as mg                                   # noqa: E402

SETTINGS = dict(range_height=256, range_width=256, random_range_crop=False, range_object_norm=True, range_object_norm_scale=0.75,
                range_int_norm=True, expand_mask_ratio=0.1, prob_drop_context=0.0)


def main():
    root = tempfile.mkdtemp(prefix="mini_db_golden_")
    csv, pkl = mini_db.build(root, n_scenes=2, seed=0)
    mg.import_reference()
    for name in ("albumentations", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    tt = sys.modules["torchvision.transforms"]
    for n in ("Compose", "ToTensor", "Normalize"):
        if not hasattr(tt, n):
            setattr(tt, n, lambda *a, **k: None)
    import ldm.data.nuscenes as nd
    nd.get_tensor = lambda normalize=True, toTensor=True: (lambda a: torch.from_numpy(np.asarray(a))[None])
    import cv2 as cv2_stub

    def nearest(img, size, interpolation=None):
        w, h = size
        ys = np.minimum((np.arange(h) * (img.shape[0] / h)).astype(np.int64), img.shape[0] - 1)
        xs = np.minimum((np.arange(w) * (img.shape[1] / w)).astype(np.int64), img.shape[1] - 1)
        return img[ys][:, xs]
    cv2_stub.resize, cv2_stub.INTER_NEAREST = nearest, 0
    nd.get_range_inpaint_mask = lambda bbox, h, w, *a, **k: torch.ones(h, w)
    with open(pkl, "rb") as f:
        scenes = pickle.load(f)
    fake = types.SimpleNamespace(**SETTINGS)
    out = {}
    for token, scene in sorted(scenes.items()):
        for k in range(len(scene["gt_bboxes_3d_corners"])):
            item = nd.NuScenesDataset.get_range_data(fake, scene, scene["gt_bboxes_3d_corners"][k], k)
            tag = f"{token}.{k}"
            for key in ("range_depth_orig", "range_int_orig", "range_instance_mask_orig", "range_pitch", "range_yaw",
                        "min_depth_obj", "max_depth_obj"):
                v = item[key]
                v = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
                if v.size > 64:                 # the untouched sweep arrays: shape + float64 sum + a strided sample (file size)
                    out[f"{tag}.{key}.shape"] = np.asarray(v.shape)
                    out[f"{tag}.{key}.sum"] = np.asarray(v.astype(np.float64).sum())
                    out[f"{tag}.{key}.sample"] = v.reshape(-1)[::97].copy()
                else:
                    out[f"{tag}.{key}"] = v
            out[f"{tag}.range_shift_left"] = np.asarray(item["range_shift_left"])
            out[f"{tag}.width_crop"] = np.asarray(item["width_crop"])
            out[f"{tag}.ref_bbox"] = item["cond"]["ref_bbox"].numpy()
    out["settings"] = np.array(repr(sorted(SETTINGS.items())))
    mg.save("data_item", **out)


if __name__ == "__main__":
    main()
