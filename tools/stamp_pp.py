#!/usr/bin/env python3
"""Per-phase shader-cycle stamps of the ping-pong igemm kernel (rebuilds the library with -DMOBI_STAMP=3).

Per wave and k-step (ks 0 / 1): cycles waiting at the LOAD barrier, in the LOAD phase (fragment reads, DMA requests,
waits), waiting at the MATRIX barrier, issuing the MATRIX phase's MFMAs.   python tools/stamp_pp.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (images, hw, cin, cout, k, geglu, residual)
    (16, 64, 320, 320, 3, False, True), (16, 32, 640, 640, 3, False, True), (16, 64, 320, 320, 1, False, True),
    (16, 64, 1280, 320, 1, False, True), (16, 64, 320, 1280, 1, True, False),
]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=3").strip()
    if os.environ.get("MOBI_PP_PHASES"):
        os.environ["MOBI_HIPCC_FLAGS"] += " -DMOBI_PP_PHASES=" + os.environ["MOBI_PP_PHASES"]
    from mobi_amd import build
    build.build(force=True, verbose=False)
    from mobi_amd import _lib, ops
    lib = _lib.load()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    cap = 1 << 12
    lib.mobi_debug_set_phases.argtypes = [C.c_void_p]
    lib.mobi_debug_set_phases.restype = C.c_int
    phases = torch.zeros(cap * 8 * 16, dtype=torch.int64, device="cuda")
    for images, hw, cin, cout, k, geglu, resid in SHAPES:
        x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
        if geglu:
            pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.zeros(2 * cout), dt, "cuda")
        else:
            pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
        res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt) if resid else None
        assert lib.mobi_debug_set_phases(None) == 0
        for _ in range(10):
            ops.igemm(x, pw, residual=res)
        torch.cuda.synchronize()
        phases.zero_()
        assert lib.mobi_debug_set_phases(C.c_void_p(phases.data_ptr())) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.igemm(x, pw, residual=res)
        e1.record()
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_phases(None) == 0
        ph = phases.cpu().numpy().reshape(cap, 8, 16)
        ph = ph[ph[:, 0, 8] != 0].astype(np.float64)
        per = ph[:, :, :8] / ph[:, :, 8:9]                  # cycles per k-tile and slot, per wave
        pm = per.mean(axis=0)                               # [wave][slot]
        names = ["ks0:barL", "load", "barM", "mfma", "ks1:barL", "load", "barM", "mfma"]
        if os.environ.get("MOBI_PP_PHASES") == "2":
            names = ["barL", "load", "barM", "mfma", "vmwait(L)", "vmwait(E)", "-", "-"]
        print(f"m={images * hw * hw} n={pw.n_packed} k={k * k * cin} geglu={int(geglu)}: launch {e0.elapsed_time(e1) * 1e3:.1f} us, "
              f"{len(ph)} blocks; cycles per k-tile (sum {pm.sum(axis=1).mean():.0f})")
        print("    slot      " + " ".join(f"{n:>9s}" for n in names))
        print("    waves 0-3 " + " ".join(f"{v:9.0f}" for v in pm[:4].mean(axis=0)))
        print("    waves 4-7 " + " ".join(f"{v:9.0f}" for v in pm[4:].mean(axis=0)), flush=True)


if __name__ == "__main__":
    main()
