#!/usr/bin/env python3
"""Where the ping-pong kernel's wrong outputs sit (per output tile, per pixel row inside a 16-row MFMA tile, per
channel inside a 16-channel tile, per component)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOBI_IGEMM_WM"] = "4"
from mobi_amd import build, ops
build.build(verbose=False)
g = torch.Generator().manual_seed(1)
dt = torch.bfloat16
cases = []
for cin in (320, 640, 704, 768, 832, 896, 960):
    cases.append((3, 16, 16, cin, 160, 1, False, False, True, 0))
cases += [(6, 16, 16, 320, 160, 1, False, False, True, 2)]
for (n, h, w, cin, cout, k, res, rowvec, bias, blocks) in cases:
    if blocks: os.environ["MOBI_IGEMM_PERSIST_BLOCKS"] = str(blocks)
    else: os.environ.pop("MOBI_IGEMM_PERSIST_BLOCKS", None)
    x = torch.randn(n, h, w, cin, generator=g).to(dt)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).to(dt)
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, h, w, cout, generator=g).to(dt) if res else None
    pw = ops.pack_conv(wt.float(), b, dt, "cuda")
    y = ops.igemm(x.cuda(), pw, residual=None if r is None else r.cuda())
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), b, padding=k // 2).permute(0, 2, 3, 1)
    if r is not None: ref = ref + r.float()
    d = (y.float().cpu() - ref).reshape(-1, cout)
    bad = (d.abs() > 0.05 * ref.abs().max()) | ~torch.isfinite(d)
    idx = bad.nonzero()
    print(f"n={n} {h}x{w} {cin}->{cout} nk={cin * k * k // 64} res={res} bias={bias} blocks={blocks}: bad={len(idx)} of {d.numel()}", flush=True)
    if len(idx):
        m, c = idx[:, 0], idx[:, 1]
        print("   per 256-px tile:", torch.bincount(m // 256).tolist())
        print("   per 64-px wave row (wm):", torch.bincount((m % 256) // 64, minlength=4).tolist())
        print("   per 16-px MFMA tile (mi):", torch.bincount((m % 64) // 16, minlength=4).tolist())
        print("   per pixel in MFMA tile (r16):", torch.bincount(m % 16, minlength=16).tolist())
        print("   per channel in 16 (4*g4+r):", torch.bincount(c % 16, minlength=16).tolist())
        print("   per 16-channel tile:", torch.bincount(c // 16).tolist())
        yy = y.float().cpu().reshape(-1, cout)
        rr = ref.reshape(-1, cout)
        for (mm, cc) in idx[:6].tolist():
            print(f"      m={mm} c={cc}: got {yy[mm, cc].item():.5f} expected {rr[mm, cc].item():.5f}; neighbours got {[round(v, 4) for v in yy[mm, cc - cc % 8: cc - cc % 8 + 8].tolist()]} exp {[round(v, 4) for v in rr[mm, cc - cc % 8: cc - cc % 8 + 8].tolist()]}")
