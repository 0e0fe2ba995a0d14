#!/usr/bin/env python3
"""LayerNorm on the shapes of one denoising step, graph-timed on rotating (cold) inputs (see tools/gn_lab.py).

    python tools/ln_lab.py [--dtype bf16] [--iters 24]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gn_lab import timeit  # noqa: E402

SHAPES = [(16, 4096, 320), (8, 4096, 320), (16, 1024, 640), (8, 1024, 640), (16, 256, 1280), (8, 256, 1280), (16, 64, 1280)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=24)
    a = ap.parse_args()
    from mobi_amd import build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    for n, t, c in SHAPES:
        mb = n * t * c * 2 / 1e6
        copies = max(2, min(12, int(600 / mb)))
        xs = [(torch.randn(n, t, c, generator=g) * 1.5 + 0.3).to("cuda").to(dt) for _ in range(2)]
        xs += [xs[i % 2].clone() for i in range(copies - 2)]
        gam, bet = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        fns = [lambda x=x: ops.layernorm(x, gam, bet, 1e-5) for x in xs]
        us = min(timeit(fns, a.iters) for _ in range(3))
        print(f"layernorm images={n:3d} tokens={t:5d} C={c:5d} {mb:6.1f} MB: {us:6.1f} us {2 * mb / us * 1e3:5.0f} GB/s")


if __name__ == "__main__":
    main()
