#!/usr/bin/env python3
"""k-order / window-addressing A/B of the direct-to-LDS kernel on 3x3 shapes (one process, interleaved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_variant import timeit  # noqa: E402

SHAPES = [(16, 64, 320, 320), (16, 32, 640, 640), (16, 64, 640, 320), (16, 32, 1280, 640), (16, 32, 1280, 1280)]


def main():
    from mobi_amd import build, ops
    build.build(verbose=False)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    for images, hw, cin, cout in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
        packs = {False: ops.pack_conv(w, torch.zeros(cout), dt, "cuda"),
                 True: ops.pack_conv(w, torch.zeros(cout), dt, "cuda", chunk_major=True)}
        fl = 2.0 * images * hw * hw * cout * cin * 9
        cells = []
        for rep in range(3):
            for cm in (False, True):
                for lin in ("1", "0"):
                    os.environ["MOBI_IGEMM_LIN"] = lin
                    us = timeit(lambda: ops.igemm(x, packs[cm]), 20)
                    cells.append(f"{'chunk' if cm else 'tap'}{'' if lin == '1' else '-nolin'}={us:6.1f}")
        os.environ.pop("MOBI_IGEMM_LIN", None)
        print(f"m={images * hw * hw} cin={cin} cout={cout} GF={fl / 1e9:.0f} | " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
