#!/usr/bin/env python3
"""Where a launch of the ring kernel spends its time (rebuilds with -DMOBI_STAMP=1): per block s_memrealtime stamps at
kernel entry, first k-step landed, end of the k loop, after the epilogue's stores; and how many blocks a CU holds over
the launch.   python tools/stamp_ring.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (images, hw, cin, cout, k, geglu, residual)
SHAPES = [(16, 64, 320, 1280, 1, True, False), (16, 32, 640, 2560, 1, True, False), (16, 16, 1280, 5120, 1, True, False),
          (16, 64, 320, 320, 1, False, True), (16, 64, 1280, 320, 1, False, True), (16, 64, 320, 320, 3, False, True),
          # small m (mobi_nusc_256's launches, the 128 x 160 tiles)
          (8, 32, 320, 320, 1, False, True), (8, 16, 640, 640, 1, False, True), (8, 8, 1280, 1280, 1, False, True),
          (8, 16, 640, 640, 3, False, True), (8, 8, 1280, 1280, 3, False, True)]


def main():
    os.environ["MOBI_HIPCC_FLAGS"] = (os.environ.get("MOBI_HIPCC_FLAGS", "") + " -DMOBI_STAMP=1").strip()
    os.environ["MOBI_IGEMM_SM_DIRECT"] = os.environ["MOBI_IGEMM_RING_DIRECT"] = "0"      # the stamped (LDS-staged) epilogue
    from mobi_amd import build
    build.build(force=True, verbose=False)
    from mobi_amd import _lib, ops
    lib = _lib.load()
    lib.mobi_debug_set_stamps.argtypes = [C.c_void_p]
    lib.mobi_debug_set_stamps.restype = C.c_int
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    cap = 1 << 13
    stamps = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
    for images, hw, cin, cout, k, geglu, resid in SHAPES:
        if geglu:
            x = torch.randn(1, images * hw * hw, cin, generator=g).cuda().to(dt)
            pw = ops.pack_geglu(torch.randn(2 * cout, cin, generator=g) / cin ** 0.5, torch.zeros(2 * cout), dt, "cuda")
            run = lambda: ops.linear(x, pw)
        else:
            x = torch.randn(images, hw, hw, cin, generator=g).cuda().to(dt)
            pw = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5), torch.zeros(cout), dt, "cuda")
            res = torch.randn(images, hw, hw, cout, generator=g).cuda().to(dt) if resid else None
            run = lambda: ops.igemm(x, pw, residual=res)
        assert lib.mobi_debug_set_stamps(None) == 0
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(20):
            run()
        e[1].record()
        torch.cuda.synchronize()
        b2b = e[0].elapsed_time(e[1]) * 1e3 / 20
        stamps.zero_()
        assert lib.mobi_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        assert lib.mobi_debug_set_stamps(None) == 0
        s = stamps.cpu().numpy().reshape(cap, 8)
        s = s[s[:, 0] != 0]
        t = s[:, :4].astype(np.float64) * 0.01
        t0 = t[:, 0].min()
        hw_id = s[:, 4]
        cu = ((hw_id >> 32) << 16) | (((hw_id >> 13) & 7) << 8) | ((hw_id >> 8) & 15)      # (xcc, se, cu)
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        # gap between one block's end and the next block's entry on the same CU
        gaps = []
        for c in np.unique(cu):
            rows = t[cu == c]
            rows = rows[np.argsort(rows[:, 0])]
            gaps += list(rows[1:, 0] - rows[:-1, 3])
        print(f"{'geglu' if geglu else 'conv'} {cin}->{cout} k{k} m={images * hw * hw}: event-timed {b2b:.1f} us back to back; "
              f"{len(s)} blocks x {int(s[:, 5].mean())} k-steps on {len(per_cu)} CUs ({per_cu.min()}-{per_cu.max()} blocks per CU) | "
              f"in-kernel span {t[:, 3].max() - t0:.1f} us: " +
              (f"entry->first step landed {np.mean(t[:, 1] - t[:, 0]):.2f}, k loop {np.mean(t[:, 2] - t[:, 1]):.2f} "
               f"({np.mean((t[:, 2] - t[:, 1]) / s[:, 5]):.3f} per step), " if (s[:, 1] != 0).all() else
               f"entry -> end of the k loop {np.mean(t[:, 2] - t[:, 0]):.2f} ({np.mean((t[:, 2] - t[:, 0]) / s[:, 5]):.3f} per step incl. "
               f"the first landing; four-wave geometry: no stamp in the loop), ") +
              f"drain + epilogue + stores {np.mean(t[:, 3] - t[:, 2]):.2f}, block end -> next block entry on the CU "
              f"{np.mean(gaps) if gaps else 0:.2f} (median {np.median(gaps) if gaps else 0:.2f}), last entry at {t[:, 0].max() - t0:.1f}",
              flush=True)
    os.environ["MOBI_HIPCC_FLAGS"] = os.environ["MOBI_HIPCC_FLAGS"].replace("-DMOBI_STAMP=1", "").strip()
    if os.environ.get("MOBI_STAMP_NO_RESTORE") != "1":       # (on a throw-away GPU box the stamped library need not be rebuilt)
        build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
