#!/usr/bin/env python3
"""Lab: the small-m 1 x 1 launches on the tile geometries the library has (MOBI_IGEMM_WIDE / _WM / _SM64 forced), graph-timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_split import timeit  # noqa: E402

SHAPES = [(16, 16, 1280, 1280, 1), (8, 16, 1280, 1280, 1), (8, 16, 1280, 2560, 1), (16, 16, 1280, 3840, 1), (16, 32, 640, 640, 1),
          (8, 32, 640, 640, 1), (8, 32, 640, 1280, 1), (16, 32, 640, 1920, 1), (16, 16, 2560, 1280, 1), (16, 16, 5120, 1280, 1),
          (16, 64, 320, 320, 1), (8, 64, 320, 320, 1)]


def main():
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    cfgs = [("plan", {}), ("ring64", {"MOBI_IGEMM_WM": "2", "MOBI_IGEMM_SM64": "1", "MOBI_IGEMM_SMALL": "0"}),
            ("ring128x160", {"MOBI_IGEMM_WM": "2", "MOBI_IGEMM_SM64": "0", "MOBI_IGEMM_SMALL": "0"}),
            ("ring128x320", {"MOBI_IGEMM_WIDE": "1", "MOBI_IGEMM_SMALL": "0"}), ("ring256x320", {"MOBI_IGEMM_WIDE": "2", "MOBI_IGEMM_SMALL": "0"}),
            ("pingpong", {"MOBI_IGEMM_WM": "4", "MOBI_IGEMM_WIDE": "0", "MOBI_IGEMM_SMALL": "0"})]
    for images, hw, cin, cout, k in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        w = torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)
        pw = ops.pack_conv(w, torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt)
        fl = 2.0 * images * hw * hw * cout * cin * k * k
        cells = []
        for tag, env in cfgs:
            os.environ.update(env)
            _lib.load().mobi_tuning_reload()
            for s in (1, 2):
                try:
                    t = timeit(lambda: ops.igemm(x, pw, residual=res, split_k=s), 30)
                    cells.append(f"{tag}/s{s} {t:5.1f}")
                except Exception:  # noqa: BLE001
                    cells.append(f"{tag}/s{s}   -  ")
            for k_ in env:
                os.environ.pop(k_, None)
        _lib.load().mobi_tuning_reload()
        print(f"m={images * hw * hw:6d} n={cout:5d} k={cin:5d} ({fl / 1e9:5.1f} GF): " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
