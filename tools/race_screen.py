#!/usr/bin/env python3
"""Race screen of the ping-pong igemm kernel: many repeats of a few shapes (different persistent block counts,
an HBM-thrashing copy between some repeats to vary landing latencies), every output compared BIT FOR BIT with the first
and once with a torch fp32 reference.  A LDS read that beats its DMA, or a stage overwritten too early, shows up as a
rare mismatch.   python tools/race_screen.py [--repeats 150]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def screen(repeats=150):
    """Returns the number of failures (mismatching repeats + shapes off the fp32 reference)."""
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    saved = {k: os.environ.get(k) for k in ("MOBI_IGEMM_WM", "MOBI_IGEMM_PERSIST_BLOCKS")}
    os.environ["MOBI_IGEMM_WM"] = "4"
    try:
        return _screen(repeats, _lib, ops)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.load().mobi_tuning_reload()


def _screen(repeats, _lib, ops):
    g = torch.Generator().manual_seed(7)
    dt = torch.bfloat16
    trash = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    bad = 0
    for (n, h, cin, cout, k, res, blocks, halo) in [(16, 32, 640, 640, 3, True, 0, "0"), (16, 32, 640, 640, 3, True, 37, "0"),
                                                    (16, 64, 320, 320, 1, True, 0, "0"), (16, 64, 320, 320, 1, True, 61, "0"),
                                                    (16, 16, 1280, 1280, 1, False, 0, "0"), (8, 32, 640, 640, 3, True, 23, "0"),
                                                    (16, 64, 320, 1280, 1, False, 101, "0")]:
        if blocks:
            os.environ["MOBI_IGEMM_PERSIST_BLOCKS"] = str(blocks)
        else:
            os.environ.pop("MOBI_IGEMM_PERSIST_BLOCKS", None)
        _lib.load().mobi_tuning_reload()          # the library reads its A/B variables once; re-read them
        x = torch.randn(n, h, h, cin, generator=g).to(dt).cuda()
        w = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).to(dt)
        b = torch.randn(cout, generator=g)
        r = torch.randn(n, h, h, cout, generator=g).to(dt).cuda() if res else None
        pw = ops.pack_conv(w.float(), b, dt, "cuda")
        y0 = ops.igemm(x, pw, residual=r)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().cuda(), b.cuda(), padding=k // 2).permute(0, 2, 3, 1)
        if r is not None:
            ref = ref + r.float()
        err = float((y0.float() - ref).norm() / ref.norm())
        mism = 0
        for i in range(repeats):
            if i % 5 == 0:
                trash.add_(1)                                # evict L2 / MALL, busy HBM
            y = ops.igemm(x, pw, residual=r)
            if not torch.equal(y, y0):
                mism += 1
        torch.cuda.synchronize()
        bad += mism + (err > 8e-3)
        print(f"n={n} {h}x{h} {cin}->{cout} k{k} res={res} blocks={blocks} halo={halo}: rel={err:.2e} "
              f"mismatching repeats {mism}/{repeats}", flush=True)
    print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=150)
    a = ap.parse_args()
    sys.exit(1 if screen(a.repeats) else 0)


if __name__ == "__main__":
    main()
