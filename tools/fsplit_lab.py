#!/usr/bin/env python3
"""Lab: split-K finished inside the launch (mobi_igemm_params.sync) against slabs + reduce launch, per shape and split count,
graph-timed on one box (tools/sweep_split.py's shapes).

    python tools/fsplit_lab.py [--set 512|256] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_split import SHAPES, SHAPES_256, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--set", default="512", choices=["512", "256"])
    ap.add_argument("--modes", default="012", help="0: reduce launch, 1 / 2: the two in-launch finishes")
    a = ap.parse_args()
    from mobi_amd import build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(0)
    for images, hw, cin, cout, k in (SHAPES if a.set == "512" else SHAPES_256):
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        w = torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)
        pw = ops.pack_conv(w, torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt)
        nk = k * k * cin // 64
        fl = 2.0 * images * hw * hw * cout * cin * k * k
        cells = []
        best = (1e9, None)
        from mobi_amd import _lib
        for mode in a.modes:                       # u: reduce launch; D: device-coherent finish; X: same-XCD finish
            os.environ["MOBI_IGEMM_FUSED_SPLIT"] = mode
            _lib.load().mobi_tuning_reload()
            for s in ((None, 1, 2, 3, 4, 6, 8, 12, 16) if mode == "0" else (None, 2, 3, 4)):
                if s is not None and s > 1 and nk // s < 2:
                    continue
                t = timeit(lambda: ops.igemm(x, pw, residual=res, split_k=s), a.iters)
                tag = f"{'uDX'[int(mode)]}{'plan' if s is None else s}"
                cells.append(f"{tag}={t:6.1f}")
                if t < best[0]:
                    best = (t, tag)
        os.environ.pop("MOBI_IGEMM_FUSED_SPLIT", None)
        _lib.load().mobi_tuning_reload()
        print(f"m={images * hw * hw:6d} n={cout:5d} k={k * k * cin:6d} nk={nk:4d} | " + " ".join(cells) +
              f" | best {best[1]} {best[0]:.1f}us {fl / best[0] / 1e6:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
