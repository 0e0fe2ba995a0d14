#!/usr/bin/env python3
"""Dataset side timed at production geometry: a miniature database with 1600 x 900 frames and 512 x 512
views; host `__getitem__` per object (the reference's design, CPU), `raw_item` + `collate_device` (host IO / geometry,
device per-pixel work) and the two kernels alone against the HBM roofline.   python tools/bench_data_side.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mobi_amd import build, ops
    build.build(verbose=False)
    from tests import mini_db
    from mobi_amd.ldm.data.nuscenes import NuScenesDataset
    from tools.sweep_split import timeit
    root = tempfile.mkdtemp(prefix="mobi_mini_db_")
    csv, pkl = mini_db.build(root, n_scenes=4, image_hw=(900, 1600))
    R = 512
    ds = NuScenesDataset(state="test", use_lidar=True, use_camera=True, object_database_path=csv, scene_database_path=pkl,
                         expand_mask_ratio=0.1, object_area_crop=0.2, num_samples_per_class=4, fixed_sampling=True,
                         object_random_crop=False, ref_aug=False, image_height=R, image_width=R, range_height=R, range_width=R,
                         object_classes=["car", "pedestrian"], range_object_norm=True, range_int_norm=True, min_lidar_points=8)
    B = len(ds)
    ds[0]; ds.raw_item(0)
    t0 = time.perf_counter()
    host = [ds[i] for i in range(B)]
    t_host = (time.perf_counter() - t0) / B
    t0 = time.perf_counter()
    raw = [ds.raw_item(i) for i in range(B)]
    t_raw = (time.perf_counter() - t0) / B
    ds.collate_device(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch = ds.collate_device(raw)
    torch.cuda.synchronize()
    t_col = (time.perf_counter() - t0) / B
    print(f"{B} objects, {R} x {R} views, 1600 x 900 frames: host __getitem__ {t_host * 1e3:.1f} ms per object (1 core); "
          f"raw_item {t_raw * 1e3:.1f} ms + collate_device {t_col * 1e3:.2f} ms per object")
    lid = batch["lidar"]
    d0, i0, m0 = (lid[k].float().contiguous() for k in ("range_depth_orig", "range_int_orig", "range_instance_mask_orig"))
    args = (d0, i0, m0, lid["range_shift_left"], lid["width_crop"], lid["min_depth_obj"].float(), lid["max_depth_obj"].float(), lid["range_mask"])
    us = timeit(lambda: ops.range_prepare(*args, height=R, width=R, alpha=0.75, object_norm=True, int_norm=True), 20, warm=2)
    nbytes = B * R * R * 4 * (2 + 2 + 1 + 1)                     # written views + the edit mask read (sweeps: L2-resident)
    print(f"mobi_range_prepare [{B}, {R}, {R}]: {us:.1f} us, {nbytes / us / 1e3:.0f} GB/s of {8000} (algorithmic {nbytes / 1e6:.1f} MB)")
    corners = (torch.rand(B, 8, 2, device="cuda") * R).to(torch.int32)
    us = timeit(lambda: ops.box_mask(corners, R, R), 20, warm=2)
    print(f"mobi_box_mask [{B}, {R}, {R}]: {us:.1f} us, {B * R * R * 4 / us / 1e3:.0f} GB/s written (24 edge tests per pixel in fp64: ALU-bound)")
    frames = torch.randint(0, 256, (B, 900, 1600, 3), dtype=torch.uint8, device="cuda")
    cw = (torch.rand(B, 8, 2, device="cuda") * 300 + 500).to(torch.int32)
    crop = torch.tensor([[400, 300, 420, 420]] * B, dtype=torch.int32).cuda()
    inv = torch.zeros(B, dtype=torch.int32, device="cuda")
    us = timeit(lambda: ops.image_prepare(frames, cw, inv, crop, height=R, width=R), 20, warm=2)
    nb = B * R * R * 4 * 7 + B * 420 * 420 * 3
    print(f"mobi_image_prepare [{B}, 900 x 1600 -> {R} x {R}]: {us:.1f} us, {nb / us / 1e3:.0f} GB/s (algorithmic {nb / 1e6:.1f} MB)")
    us = timeit(lambda: ops.box_mask(cw, 900, 1600, want_mask=False, want_stats=True), 20, warm=2)
    print(f"mobi_box_mask stats only [{B}, 900, 1600]: {us:.1f} us")


if __name__ == "__main__":
    main()
