#!/usr/bin/env python3
"""Split-K plan sweep: times the step's small-m igemm shapes at several split factors (one process, same box).

    python tools/sweep_split.py [--dtype bf16] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES_256 = [  # BASELINE config 2 (mobi_nusc_256, batch 4 -> 8 images at a 32x32 latent): every launch is small-m
    (8, 32, 320, 320, 3), (8, 32, 320, 320, 1), (8, 32, 320, 960, 1), (4, 32, 320, 640, 1), (8, 32, 1280, 320, 1),
    (8, 16, 640, 640, 3), (8, 16, 640, 640, 1), (8, 16, 640, 1920, 1), (4, 16, 640, 1280, 1), (8, 16, 2560, 640, 1),
    (8, 8, 1280, 1280, 3), (8, 8, 1280, 1280, 1), (8, 8, 1280, 3840, 1), (4, 8, 1280, 2560, 1), (8, 8, 5120, 1280, 1),
    (8, 4, 1280, 1280, 3), (8, 4, 2560, 1280, 3), (8, 8, 2560, 1280, 3), (8, 16, 1280, 640, 3), (8, 32, 640, 320, 3),
    (8, 32, 960, 320, 3),
]
SHAPES = [  # (images, hw, cin, cout, k)
    (16, 16, 1280, 1280, 1), (8, 16, 1280, 1280, 1), (16, 8, 1280, 1280, 1), (8, 8, 1280, 1280, 1),
    (16, 16, 5120, 1280, 1), (16, 16, 2560, 1280, 1), (16, 8, 2560, 1280, 1), (16, 8, 5120, 1280, 1),
    (16, 16, 1280, 1280, 3), (16, 16, 2560, 1280, 3), (16, 8, 1280, 1280, 3), (16, 8, 2560, 1280, 3),
    (16, 16, 640, 1280, 3), (16, 32, 320, 640, 3), (16, 32, 640, 640, 1), (8, 32, 640, 640, 1),
    (16, 32, 640, 640, 3), (16, 32, 1280, 640, 3),
]


def timeit(fn, iters, warm=3):
    """GPU time per launch: the launches are captured in a HIP graph and replayed, so short kernels are not hidden
    behind the ~15 us a Python + ctypes call costs the host."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--set", default="512", choices=["512", "256"])
    a = ap.parse_args()
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    reload_ = _lib.load().mobi_tuning_reload
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    for images, hw, cin, cout, k in (SHAPES if a.set == "512" else SHAPES_256):
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        w = torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)
        pw = ops.pack_conv(w, torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt)
        nk = k * k * cin // 64
        fl = 2.0 * images * hw * hw * cout * cin * k * k
        line = f"m={images * hw * hw:6d} n={cout:5d} k={k * k * cin:6d} nk={nk:4d} planned={'-':>2s}:"
        cells = []
        planned = timeit(lambda: ops.igemm(x, pw, residual=res), a.iters)
        planned = timeit(lambda: ops.igemm(x, pw, residual=res), a.iters)
        # s = LDS-DMA ring kernel (128-pixel tiles), r = register-staged kernel (128), g = 256-pixel ping-pong slabs
        for tag, env in (("s", {"MOBI_IGEMM_WM": "2"}), ("r", {"MOBI_IGEMM_WM": "2", "MOBI_IGEMM_SM": "0"}),
                         ("g", {"MOBI_IGEMM_WM": "4"})):
            os.environ.update(env)
            reload_()
            for s in (1, 2, 3, 4, 6, 8, 12, 16):
                if s > 1 and nk // s < 2:
                    continue
                if tag == "g" and (images * hw * hw) % 256:
                    continue
                us = timeit(lambda: ops.igemm(x, pw, residual=res, split_k=s), a.iters)
                cells.append((us, s, tag))
            for k_ in env:
                os.environ.pop(k_, None)
        reload_()
        best = min(cells)
        print(f"m={images * hw * hw:6d} n={cout:5d} k={k * k * cin:6d} nk={nk:4d} plan={planned:7.1f}us | " +
              " ".join(f"{tag}{s}={us:5.1f}" for us, s, tag in cells) +
              f" | best {best[2]} s{best[1]} {best[0]:6.1f}us {fl / best[0] / 1e6:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
