#!/usr/bin/env python3
"""Where a k-step of the eight-wave 256 x 320 ring tiles goes (rebuilds with -DMOBI_STAMP=4): shader cycles per wave and
step, early (waves 0-3) and late (4-7) halves apart:  wait at the LOAD barrier | fragment reads + requests issued | LDS
wait | counted wait (late) | wait at the MATRIX barrier | 40 MFMAs | counted wait (early).   python tools/phase_ring256.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(16, 64, 320, 320, 3, False), (16, 64, 960, 320, 3, False), (16, 64, 1280, 320, 1, False), (16, 64, 320, 1280, 1, True)]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=4").strip()
    os.environ["MOBI_IGEMM_WIDE"] = "2"
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
    names = ["LOAD barrier", "reads + requests issued", "LDS wait", "counted wait (late)", "MATRIX barrier", "40 MFMAs", "counted wait (early)"]
    for images, hw, cin, cout, k, geglu in SHAPES:
        if geglu:
            x = torch.randn(1, images * hw * hw, cin, generator=g).cuda().to(dt)
            pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.zeros(2 * cout), dt, "cuda")
            run = lambda: ops.linear(x, pw)
        else:
            x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
            pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
            run = lambda: ops.igemm(x, pw)
        assert lib.mobi_debug_set_phases(None) == 0
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.mobi_debug_set_phases(C.c_void_p(buf.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_phases(None) == 0
        d = buf.cpu().numpy().reshape(cap, 8, 8)
        d = d[d[:, 0, 7] > 0]
        for half, sl in (("early", slice(0, 4)), ("late", slice(4, 8))):
            w = d[:, sl, :].reshape(-1, 8)
            per = (w[:, :7] / w[:, 7:8]).mean(0)
            print(f"{'geglu' if geglu else 'conv'} {cin}->{cout} k{k} m={images * hw * hw} {half:5s}: " +
                  " | ".join(f"{n} {v:.0f}" for n, v in zip(names, per)) + f" | sum {per.sum():.0f}", flush=True)
    os.environ["MOBI_HIPCC_FLAGS"] = os.environ["MOBI_HIPCC_FLAGS"].replace("-DMOBI_STAMP=4", "").strip()
    build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
