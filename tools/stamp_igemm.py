#!/usr/bin/env python3
"""In-kernel phase stamps of the direct-to-LDS igemm kernel (rebuilds the library with -DMOBI_STAMP=1 first).

Per shape: kernel span, and per block the mean time from entry to 'first tile landed', of the k loop (per k-tile),
of the epilogue (incl. the wait for its stores), and the idle gap between consecutive blocks of one CU.

    python tools/stamp_igemm.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (images, hw, cin, cout, k, geglu, residual)
    (16, 64, 320, 320, 1, False, True), (16, 64, 320, 1280, 1, True, False), (16, 64, 1280, 320, 1, False, True),
    (16, 32, 640, 640, 1, False, True), (16, 32, 640, 2560, 1, True, False),
    (16, 64, 320, 320, 3, False, True), (16, 32, 640, 640, 3, False, True), (16, 64, 640, 320, 3, False, True),
    (16, 32, 1280, 640, 3, False, True), (16, 32, 1280, 1280, 3, False, True),
]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=2").strip()
    from mobi_amd import build
    build.build(force=True, verbose=False)
    from mobi_amd import _lib, ops
    lib = _lib.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    cap = 1 << 16
    stamps = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
    lib.mobi_debug_set_stamps.argtypes = [C.c_void_p]
    lib.mobi_debug_set_stamps.restype = C.c_int
    lib.mobi_debug_set_phases.argtypes = [C.c_void_p]
    lib.mobi_debug_set_phases.restype = C.c_int
    phases = torch.zeros(cap * 64, dtype=torch.int64, device="cuda")
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
        stamps.zero_()
        phases.zero_()
        assert lib.mobi_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
        assert lib.mobi_debug_set_phases(C.c_void_p(phases.data_ptr())) == 0
        ops.igemm(x, pw, residual=res)
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_stamps(None) == 0
        assert lib.mobi_debug_set_phases(None) == 0
        ph = phases.cpu().numpy().reshape(cap, 8, 8)
        ph = ph[ph[:, 0, 5] != 0].astype(np.float64)
        per = ph[:, :, :5] / ph[:, :, 5:6] * 0.01           # us per k-tile step and phase, per wave
        pm = per.mean(axis=0)                               # [wave][phase]
        print("    phases us/k-tile (wait, barrier, k-step0, dma-issue, k-step1): all waves " +
              " ".join(f"{v:5.3f}" for v in pm.mean(axis=0)) + f" sum={pm.mean(axis=0).sum():5.3f} | wave0 " +
              " ".join(f"{v:5.3f}" for v in pm[0]) + " | wave7 " + " ".join(f"{v:5.3f}" for v in pm[7]))
        s = stamps.cpu().numpy().reshape(cap, 8)
        s = s[s[:, 0] != 0]
        t = s[:, :4].astype(np.float64) * 0.01          # 100 MHz -> us
        nk = s[:, 5]
        rounds = np.maximum(s[:, 6], 1)                 # output tiles walked by the (persistent) block
        span = t[:, 3].max() - t[:, 0].min()
        fl = 2.0 * images * hw * hw * pw.n_packed * cin * k * k
        per_tile = (t[:, 3] - t[:, 1]) / rounds
        print(f"m={images * hw * hw:6d} n={pw.n_packed:5d} k={k * k * cin:6d} geglu={int(geglu)} blocks={len(s):5d} "
              f"tiles/block={rounds.mean():4.1f} span={span:7.1f}us ({fl / span / 1e6:5.0f} TF) | "
              f"first-tile={np.mean(t[:, 1] - t[:, 0]):5.2f} per-output-tile={per_tile.mean():6.2f} (nk={int(nk.mean())}: "
              f"{np.mean(per_tile / nk):5.3f}/k-tile incl. epilogue) last-epilogue={np.mean(t[:, 3] - t[:, 2]):5.2f} "
              f"block={np.mean(t[:, 3] - t[:, 0]):6.2f} start-skew={t[:, 0].max() - t[:, 0].min():5.2f} "
              f"end-skew={t[:, 3].max() - t[:, 3].min():5.2f}", flush=True)


if __name__ == "__main__":
    main()
