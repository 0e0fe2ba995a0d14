#!/usr/bin/env python3
"""A/B of the small-m main loops on mobi_nusc_256's launches: 64-deep steps (igemm_ring64_kernel) against the 32-deep ring
with request-image weights and with row-segment weights; graph-timed microseconds, interleaved, best of three.
    python tools/ab_sm64.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_split import timeit                       # noqa: E402

CONVS = [  # (images, hw, cin, cout, k)
    (8, 32, 320, 320, 1), (8, 32, 320, 320, 3), (8, 16, 640, 640, 1), (8, 16, 640, 640, 3), (8, 16, 1280, 640, 3),
    (8, 8, 1280, 1280, 1), (8, 8, 1280, 1280, 3), (8, 8, 2560, 1280, 3), (8, 4, 1280, 1280, 3), (4, 16, 640, 640, 1),
    (16, 16, 1280, 1280, 1), (16, 8, 1280, 1280, 3), (8, 8, 5120, 1280, 1),
]


def main():
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    reload_ = _lib.load().mobi_tuning_reload
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    variants = (("k64", {}), ("k32 images", {"MOBI_IGEMM_SM64": "0"}), ("k32 rows", {"MOBI_IGEMM_SM64": "0", "MOBI_IGEMM_WTILED": "0"}))
    for images, hw, cin, cout, k in CONVS:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.randn(cout, generator=g) * 0.1, dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt)
        fn = lambda: ops.igemm(x, pw, residual=res)
        best, outs = {}, {}
        for rep in range(3):
            for tag, env in variants:
                os.environ.update(env)
                reload_()
                if rep == 0:
                    outs[tag] = fn().float()
                best[tag] = min(best.get(tag, 1e30), timeit(fn, 10, warm=1))
                for k_ in env:
                    os.environ.pop(k_, None)
        reload_()
        d = max(float((outs["k32 rows"] - o).abs().max()) for o in outs.values())
        fl = 2.0 * images * hw * hw * cout * cin * k * k
        print(f"conv {cin}->{cout} k{k} {hw}x{hw}x{images} (m={images * hw * hw}): " +
              " | ".join(f"{t} {best[t]:6.1f} us {fl / best[t] / 1e6:5.0f}" for t, _ in variants) + f" | max diff {d:.3g}", flush=True)


if __name__ == "__main__":
    main()
