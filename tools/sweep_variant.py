#!/usr/bin/env python3
"""Main-loop variant sweep (one process, same box): direct-to-LDS 256-px kernel vs register-staged 256-px / 128-px
kernels on the step's 1x1 and 3x3 shapes.  The variant is chosen per call through the MOBI_IGEMM_* overrides.

    python tools/sweep_variant.py [--dtype bf16] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (rows or (images, hw), cin, cout, k, geglu, residual)
SHAPES = [
    ((16, 64), 320, 320, 1, False, True), ((16, 64), 320, 640, 1, False, False), ((16, 64), 320, 1280, 1, True, False),
    ((16, 64), 1280, 320, 1, False, True), ((8, 64), 320, 320, 1, False, True),
    ((16, 32), 640, 640, 1, False, True), ((16, 32), 640, 2560, 1, True, False), ((16, 32), 2560, 640, 1, False, True),
    ((16, 32), 640, 1280, 1, False, False), ((8, 32), 640, 640, 1, False, True),
    ((16, 16), 1280, 5120, 1, True, False), ((16, 16), 1280, 2560, 1, False, False),
    ((16, 64), 320, 320, 3, False, True), ((16, 32), 640, 640, 3, False, True), ((16, 64), 640, 320, 3, False, True),
    ((16, 32), 1280, 640, 3, False, True),
]
VARIANTS = [("glds", {}), ("reg256", {"MOBI_IGEMM_GLDS": "0", "MOBI_IGEMM_WM": "4"}),
            ("reg128", {"MOBI_IGEMM_GLDS": "0", "MOBI_IGEMM_WM": "2"})]


def timeit(fn, iters, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from mobi_amd import build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    for (images, hw), cin, cout, k, geglu, resid in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        if geglu:
            pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.zeros(2 * cout), dt, "cuda")
        else:
            pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt) if resid else None
        fl = 2.0 * images * hw * hw * pw.n_packed * cin * k * k
        cells = []
        for rep in range(2):
            for name, env in VARIANTS:
                for key in ("MOBI_IGEMM_GLDS", "MOBI_IGEMM_WM"):
                    os.environ.pop(key, None)
                os.environ.update(env)
                us = timeit(lambda: ops.igemm(x, pw, residual=res), a.iters)
                cells.append(f"{name}={us:6.1f}")
        for key in ("MOBI_IGEMM_GLDS", "MOBI_IGEMM_WM"):
            os.environ.pop(key, None)
        print(f"m={images * hw * hw:6d} n={pw.n_packed:5d} k={k * k * cin:6d} geglu={int(geglu)} res={int(resid)} "
              f"GF={fl / 1e9:6.1f} | " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
