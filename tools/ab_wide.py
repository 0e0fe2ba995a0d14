#!/usr/bin/env python3
"""A/B of the 256 x 320 ring kernel (MOBI_IGEMM_WIDE=1) against the default routing on the step's large launches:
graph-timed GPU microseconds per launch and the largest output difference (same inputs).
    python tools/ab_wide.py [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_split import timeit                       # noqa: E402

CONVS = [  # (images, hw, cin, cin2, cout, k, residual)
    (16, 64, 320, 0, 320, 3, True), (16, 64, 640, 0, 320, 3, True), (16, 64, 960, 0, 320, 3, False),
    (16, 32, 640, 0, 640, 3, True), (16, 32, 1280, 0, 640, 3, True), (16, 32, 320, 0, 640, 3, False),
    (16, 16, 1280, 0, 1280, 3, True), (16, 64, 320, 0, 320, 1, True), (16, 64, 320, 0, 960, 1, False),
    (16, 64, 1280, 0, 320, 1, True), (16, 32, 640, 0, 1920, 1, False), (16, 32, 2560, 0, 640, 1, True),
    (8, 64, 320, 0, 320, 1, True), (16, 32, 640, 0, 640, 1, True), (8, 32, 640, 0, 640, 1, True),
    (16, 16, 1280, 0, 1280, 1, True), (8, 16, 1280, 0, 1280, 1, True), (16, 16, 1280, 0, 3840, 1, False),
]
GEGLU = [(65536, 320, 1280), (16384, 640, 2560), (4096, 1280, 5120)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    reload_ = _lib.load().mobi_tuning_reload
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)

    def ab(fn, fl, name):
        # variants are timed INTERLEAVED, three rounds, best of each: the first kernel timed after a pause runs ~10 %
        # slower (clocks), which an A-then-B order books to A
        variants = (("routed", {}), ("pp", {"MOBI_IGEMM_WIDE": "0"}), ("w256", {"MOBI_IGEMM_WIDE": "2"}),
                    ("w256 row-segment weights", {"MOBI_IGEMM_WIDE": "2", "MOBI_IGEMM_WTILED": "0"}),
                    ("w256 staged epi", {"MOBI_IGEMM_WIDE": "2", "MOBI_IGEMM_RING_DIRECT": "0"}),
                    ("w128", {"MOBI_IGEMM_WIDE": "1"}), ("sm", {"MOBI_IGEMM_WM": "2", "MOBI_IGEMM_WIDE": "0"}),
                    ("sm row-segment weights", {"MOBI_IGEMM_WM": "2", "MOBI_IGEMM_WIDE": "0", "MOBI_IGEMM_WTILED": "0"}))
        best, outs = {}, {}
        for rep in range(3):
            for tag, env in variants:
                os.environ.update(env)
                reload_()
                if rep == 0:
                    outs[tag] = fn().float()
                us = timeit(fn, a.iters, warm=1)
                best[tag] = min(best.get(tag, 1e30), us)
                for k_ in env:
                    os.environ.pop(k_, None)
        reload_()
        d = max(float((outs["pp"] - outs[t]).abs().max()) for t in outs)
        print(f"{name:46s} " + " | ".join(f"{t} {best[t]:6.1f} us {fl / best[t] / 1e6:5.0f}" for t, _ in variants) +
              f" | max diff {d:.3g}", flush=True)

    for images, hw, cin, cin2, cout, k, resid in CONVS:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        w = torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)
        pw = ops.pack_conv(w, torch.randn(cout, generator=g) * 0.1, dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt) if resid else None
        ab(lambda: ops.igemm(x, pw, residual=res), 2.0 * images * hw * hw * cout * cin * k * k,
           f"conv {cin}->{cout} k{k} {hw}x{hw}x{images} res={resid}")
    for rows, cin, cout in GEGLU:
        x = torch.randn(1, rows, cin, generator=g).cuda().to(dt)
        pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.randn(2 * cout, generator=g) * 0.1, dt, "cuda")
        ab(lambda: ops.linear(x, pw), 2.0 * rows * cin * 2 * cout, f"geglu {cin}->{cout} rows={rows}")


if __name__ == "__main__":
    main()
