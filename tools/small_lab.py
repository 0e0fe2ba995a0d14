#!/usr/bin/env python3
"""A/B of the small-problem kernel (csrc/igemm_small.hip) against the LDS-ring kernels (+ their split-K reduce
launch) on the 1 x 1 launches of both steps: graph-timed microseconds per launch on COLD operands (the launches of a graph
rotate through enough copies of weights and activations to exceed the 256-MB Infinity Cache), interleaved, best of three.
    python tools/small_lab.py [--set 256|512|all]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (rows, cin, cout, residual) -- the transformer blocks' linears and the ResBlock skip convolutions
SHAPES_256 = [(256, 1280, 1280, True), (1024, 640, 640, True), (512, 1280, 1280, True), (4096, 320, 320, True),
              (2048, 640, 640, True), (8192, 320, 320, True), (256, 1280, 2560, False), (1024, 640, 1280, False),
              (4096, 320, 640, False), (8192, 320, 960, False), (2048, 640, 1920, False), (512, 1280, 3840, False),
              (64, 1280, 1280, True), (128, 1280, 1280, True), (128, 2560, 1280, False), (512, 2560, 1280, False),
              (2048, 2560, 640, False), (512, 5120, 1280, True), (8192, 1280, 320, True), (2048, 2560, 640, True)]
SHAPES_512 = [(2048, 1280, 1280, True), (4096, 1280, 1280, True), (8192, 640, 640, True), (16384, 640, 640, True),
              (2048, 1280, 2560, False), (8192, 640, 1280, False), (1024, 1280, 1280, True), (512, 1280, 1280, True),
              (4096, 1280, 3840, False), (16384, 640, 1920, False), (32768, 320, 320, True), (1024, 2560, 1280, False),
              (4096, 2560, 1280, False), (4096, 5120, 1280, True), (16384, 2560, 640, True), (65536, 320, 320, True)]


# (images, side, cin, cout): 3 x 3 convolutions with few pixels (the 4 x 4 / 8 x 8 levels)
CONVS = [(8, 4, 1280, 1280), (8, 4, 2560, 1280), (16, 4, 1280, 1280), (8, 8, 1280, 1280), (16, 8, 1280, 1280), (8, 8, 2560, 1280)]


COPIES = 0


def graph_time(fns, reps=2):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


def conv_lab(ops, reload_, dt, gen):
    variants = (("ring", {"MOBI_IGEMM_SMALL": "0"}), ("small", {"MOBI_IGEMM_SMALL": "32"}))
    for images, side, cin, cout in CONVS:
        copies = COPIES if COPIES > 0 else max(4, min(24, int(400e6 / (cout * cin * 9 * 2)) + 1))
        w0 = torch.randn(cout, cin, 3, 3, generator=gen) / (3 * cin ** 0.5)
        b0 = torch.randn(cout, generator=gen) * 0.1
        pws = [ops.pack_conv(w0, b0, dt, "cuda") for _ in range(copies)]
        xs = [torch.randn(images, side, side, cin, generator=gen).cuda().to(dt) for _ in range(copies)]
        rv = torch.randn(images, cout, generator=gen).cuda()
        fns = [(lambda i=i: ops.igemm(xs[i], pws[i], rowvec=rv)) for i in range(copies)] * (8 if copies == 1 else 1)
        best, first = {}, {}
        for rep in range(3):
            for tag, env in variants:
                os.environ.update(env)
                reload_()
                if rep == 0:
                    first[tag] = fns[0]().float().clone()
                best[tag] = min(best.get(tag, 1e30), graph_time(fns))
                for k_ in env:
                    os.environ.pop(k_, None)
        reload_()
        d = float((first["ring"] - first["small"]).abs().max())
        fl = 2.0 * images * side * side * cin * 9 * cout
        print(f"conv3x3 {cin}->{cout} {side}x{side}x{images} (m={images * side * side}) {fl / 1e9:6.2f} GF: " +
              " | ".join(f"{t} {best[t]:6.1f} us {fl / best[t] / 1e6:5.0f} TF/s" for t, _ in variants) + f" | max diff {d:.3g}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="all")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--copies", type=int, default=0, help="operand copies a graph rotates through (0: enough to exceed the Infinity Cache; 1: hot operands)")
    a = ap.parse_args()
    global COPIES
    COPIES = a.copies
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    reload_ = _lib.load().mobi_tuning_reload
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    gen = torch.Generator().manual_seed(0)
    shapes = {"256": SHAPES_256, "512": SHAPES_512, "all": SHAPES_256 + SHAPES_512, "conv": []}[a.set]
    if a.set in ("conv", "all"):
        conv_lab(ops, reload_, dt, gen)
    variants = (("ring", {"MOBI_IGEMM_SMALL": "0"}), ("small", {"MOBI_IGEMM_SMALL": "32"}))
    for rows, cin, cout, res in shapes:
        per = (rows * cin + cout * cin + rows * cout * (2 if res else 1)) * 2
        copies = COPIES if COPIES > 0 else max(4, min(96, int(400e6 / per) + 1))
        w0 = torch.randn(cout, cin, 1, 1, generator=gen) / cin ** 0.5
        b0 = torch.randn(cout, generator=gen) * 0.1
        pws = [ops.pack_conv(w0, b0, dt, "cuda") for _ in range(copies)]
        xs = [torch.randn(1, rows, 1, cin, generator=gen).cuda().to(dt) for _ in range(max(1, min(copies, 8)))]
        xs = [xs[i % len(xs)].clone() for i in range(copies)]
        rs = [torch.randn(1, rows, 1, cout, generator=gen).cuda().to(dt) if res else None for _ in range(copies)]
        outs_ = [torch.empty(1, rows, 1, cout, device="cuda", dtype=dt) for _ in range(copies)]
        fns = [(lambda i=i: ops.igemm(xs[i], pws[i], residual=rs[i], out=outs_[i])) for i in range(copies)] * (8 if copies == 1 else 1)
        best, first = {}, {}
        for rep in range(3):
            for tag, env in variants:
                os.environ.update(env)
                reload_()
                if rep == 0:
                    first[tag] = fns[0]().float().clone()
                best[tag] = min(best.get(tag, 1e30), graph_time(fns))
                for k_ in env:
                    os.environ.pop(k_, None)
        reload_()
        d = max(float((first["ring"] - o).abs().max()) for o in first.values())
        fl = 2.0 * rows * cin * cout
        print(f"m={rows:6d} k={cin:5d} n={cout:5d} {fl / 1e9:6.2f} GF res={int(res)}: " +
              " | ".join(f"{t} {best[t]:6.1f} us {fl / best[t] / 1e6:5.0f} TF/s" for t, _ in variants) + f" | max diff {d:.3g}", flush=True)


if __name__ == "__main__":
    main()
