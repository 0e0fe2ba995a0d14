#!/usr/bin/env python3
"""GroupNorm(32)+SiLU forms side by side on every shape of one denoising step: two launches (MOBI_GN_FUSED=0), one launch
with the slab in LDS (=1, where it fits), one launch with the slab in registers (default).  Interleaved best of 3 replays of a captured graph of the launches; the
inputs rotate over enough distinct tensors that no form reads its input from a cache the step would not have.

    python tools/gn_lab.py [--images 16] [--dtype bf16] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (hw, C, C of the second source, launches per mobi_nusc_512 step)
SHAPES_512 = [(4096, 320, 0, 13), (4096, 640, 320, 2), (4096, 960, 320, 1), (1024, 320, 0, 1), (1024, 640, 0, 11),
              (1024, 960, 320, 1), (1024, 1280, 640, 1), (1024, 1920, 640, 1), (256, 640, 0, 1), (256, 1280, 0, 11),
              (256, 1920, 640, 1), (256, 2560, 1280, 2), (64, 1280, 0, 12), (64, 2560, 1280, 3)]


# mobi_nusc_256 (run with --images 8)
SHAPES_256 = [(1024, 320, 0, 13), (1024, 640, 320, 2), (1024, 960, 320, 1), (256, 320, 0, 1), (256, 640, 0, 11),
              (256, 960, 320, 1), (256, 1280, 640, 1), (256, 1920, 640, 1), (64, 640, 0, 1), (64, 1280, 0, 11),
              (64, 1920, 640, 1), (64, 2560, 1280, 2), (16, 1280, 0, 12), (16, 2560, 1280, 3)]


def timeit(fns, iters):
    """Device time per launch: the rotation is captured in a HIP graph (the host cannot issue 5-us kernels fast enough)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns:
            f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            fns[i % len(fns)]()
    graph.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    del graph
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--set", default="512", choices=["512", "256"])
    a = ap.parse_args()
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    # (MOBI_GN_FUSED, MOBI_GN_COOP): the register form without the chunked kernel, and the chunked kernel wherever it fits
    forms = (("two-launch", ("0", "0")), ("lds", ("1", "0")), ("registers", (None, "0")), ("chunks", (None, "1")))
    total = {t: 0.0 for t, _ in forms}
    print(f"GroupNorm(32)+SiLU, {a.images} images, {a.dtype}; us per launch (algorithmic GB/s at 2 B read + 2 B written)")
    for hw, c, c1, n in (SHAPES_512 if a.set == "512" else SHAPES_256):
        mb = a.images * hw * c * 2 / 1e6
        copies = max(2, min(12, int(600 / mb)))               # > 256 MB of distinct inputs where that is cheap
        xs = [(torch.randn(a.images, hw, c, generator=g) * 1.5 + 0.3).to("cuda").to(dt) for _ in range(2)]
        xs += [xs[i % 2].clone() for i in range(copies - 2)]
        gam = (torch.randn(c, generator=g) * 0.2 + 1).cuda()
        bet = (torch.randn(c, generator=g) * 0.2).cuda()

        if c1:
            pairs = [(x[..., :c - c1].contiguous().view(a.images, 1, hw, c - c1),
                      x[..., c - c1:].contiguous().view(a.images, 1, hw, c1)) for x in xs]
            fns = [lambda p=p: ops.groupnorm(p[0], gam, bet, 1e-5, True, p[1]) for p in pairs]
        else:
            fns = [lambda x=x: ops.groupnorm(x.view(a.images, 1, hw, c), gam, bet, 1e-5, True) for x in xs]
        x32 = xs[0].float().view(a.images, hw, 32, c // 32)
        mu = x32.mean(dim=(1, 3), keepdim=True)
        var = x32.var(dim=(1, 3), keepdim=True, unbiased=False)
        ref = ((x32 - mu) * torch.rsqrt(var + 1e-5)).view(a.images, hw, c) * gam + bet
        ref = ref * torch.sigmoid(ref)
        best, err = {}, {}
        for rep in range(3):
            for tag, env in forms:
                for var, val in zip(("MOBI_GN_FUSED", "MOBI_GN_COOP"), env):
                    if val is None:
                        os.environ.pop(var, None)
                    else:
                        os.environ[var] = val
                _lib.load().mobi_tuning_reload()
                if rep == 0:
                    out = fns[0]().float().view(a.images, hw, c)
                    err[tag] = float((out - ref).norm() / ref.norm())
                best[tag] = min(best.get(tag, 1e30), timeit(fns, a.iters))
        os.environ.pop("MOBI_GN_FUSED", None)
        os.environ.pop("MOBI_GN_COOP", None)
        _lib.load().mobi_tuning_reload()
        for t in total:
            total[t] += best[t] * n
        print(f"hw={hw:5d} C={c:5d} ({c - c1}+{c1}) x{n:2d} {mb:6.1f} MB: " +
              " | ".join(f"{t} {best[t]:6.1f} us {2 * mb / best[t] * 1e3:5.0f} GB/s err {err[t]:.1e}" for t, _ in forms))
    print("per step (61 launches): " + " | ".join(f"{t} {v / 1e3:.3f} ms" for t, v in total.items()))


if __name__ == "__main__":
    main()
