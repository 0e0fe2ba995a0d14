#!/usr/bin/env python3
"""Where a short ping-pong launch spends its time (rebuilds with -DMOBI_STAMP=1): per block s_memrealtime stamps at
kernel entry, first k-tile landed, end of the k loops, after the last epilogue's stores.   python tools/stamp_pp_launch.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(16, 64, 320, 320, 1, False, True), (8, 64, 320, 320, 1, False, True), (16, 32, 640, 640, 1, False, True),
          (16, 64, 320, 1280, 1, True, False), (16, 64, 1280, 320, 1, False, True), (16, 64, 320, 320, 3, False, True)]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=1").strip()
    from mobi_amd import build
    build.build(force=True, verbose=False)
    from mobi_amd import _lib, ops
    lib = _lib.load()
    lib.mobi_debug_set_stamps.argtypes = [C.c_void_p]
    lib.mobi_debug_set_stamps.restype = C.c_int
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    cap = 1 << 12
    stamps = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
    for images, hw, cin, cout, k, geglu, resid in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        if geglu:
            pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.zeros(2 * cout), dt, "cuda")
        else:
            pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt) if resid else None
        assert lib.mobi_debug_set_stamps(None) == 0
        for _ in range(10):
            ops.igemm(x, pw, residual=res)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record(); ops.igemm(x, pw, residual=res); e[1].record()
        for _ in range(20):
            ops.igemm(x, pw, residual=res)
        e[2].record()
        torch.cuda.synchronize()
        single, b2b = e[0].elapsed_time(e[1]) * 1e3, e[1].elapsed_time(e[2]) * 1e3 / 20
        stamps.zero_()
        assert lib.mobi_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
        ops.igemm(x, pw, residual=res)
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_stamps(None) == 0
        s = stamps.cpu().numpy().reshape(cap, 8)
        s = s[s[:, 0] != 0]
        t = s[:, :4].astype(np.float64) * 0.01
        rounds = np.maximum(s[:, 6], 1)
        nk = s[:, 5]
        print(f"m={images * hw * hw} n={pw.n_packed} k={k * k * cin}: event-timed {b2b:.1f} us back to back ({single:.1f} alone); "
              f"{len(s)} blocks x {rounds.mean():.1f} tiles x {int(nk.mean())} k-tiles | in-kernel span {t[:, 3].max() - t[:, 0].min():.1f} us: "
              f"start skew {t[:, 0].max() - t[:, 0].min():.2f}, entry->first tile {np.mean(t[:, 1] - t[:, 0]):.2f}, "
              f"k loops {np.mean(t[:, 2] - t[:, 1]):.2f} ({np.mean((t[:, 2] - t[:, 1]) / rounds / nk):.3f} per k-tile), "
              f"last epilogue + drain {np.mean(t[:, 3] - t[:, 2]):.2f}, end skew {t[:, 3].max() - t[:, 3].min():.2f}", flush=True)
    build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
