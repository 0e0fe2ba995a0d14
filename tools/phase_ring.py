#!/usr/bin/env python3
"""Where a k-step of the four-wave (128 x 160) ring kernel goes on small launches (rebuilds with -DMOBI_STAMP=4): shader
cycles per wave and step in  counted wait | barrier | fragment reads + requests (issue) | LDS wait | 20 MFMAs.
    python tools/phase_ring.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (images, hw, cin, cout, k)
SHAPES = [(8, 32, 320, 320, 1), (8, 16, 640, 640, 1), (8, 8, 1280, 1280, 1), (8, 16, 640, 640, 3), (8, 8, 1280, 1280, 3),
          (16, 16, 1280, 1280, 1), (16, 8, 1280, 1280, 3)]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=4").strip()
    os.environ["MOBI_IGEMM_WM"] = "2"
    os.environ["MOBI_IGEMM_WIDE"] = "0"
    from mobi_amd import build
    build.build(force=True, verbose=False)
    from mobi_amd import _lib, ops
    lib = _lib.load()
    lib.mobi_debug_set_phases.argtypes = [C.c_void_p]
    lib.mobi_debug_set_phases.restype = C.c_int
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    cap = 1 << 12
    buf = torch.zeros(cap * 8 * 8, dtype=torch.int64, device="cuda")
    for images, hw, cin, cout, k in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt)
        run = lambda: ops.igemm(x, pw, residual=res)
        assert lib.mobi_debug_set_phases(None) == 0
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.mobi_debug_set_phases(C.c_void_p(buf.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_phases(None) == 0
        d = buf.cpu().numpy().reshape(cap * 8, 8)
        d = d[d[:, 5] > 0]
        per = d[:, :5] / d[:, 5:6]
        m = per.mean(0)
        print(f"conv {cin}->{cout} k{k} m={images * hw * hw}: {len(d)} waves x {int(d[:, 5].mean())} steps | cycles per step: "
              f"counted wait {m[0]:.0f} | barrier {m[1]:.0f} | fragment reads + requests issued {m[2]:.0f} | LDS wait {m[3]:.0f} | "
              f"MFMAs {m[4]:.0f} | sum {m.sum():.0f}", flush=True)
    os.environ["MOBI_HIPCC_FLAGS"] = os.environ["MOBI_HIPCC_FLAGS"].replace("-DMOBI_STAMP=4", "").strip()
    build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
