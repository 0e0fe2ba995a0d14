#!/usr/bin/env python3
"""Correctness of the ping-pong igemm kernel on a few shapes against a torch fp32 conv (GPU), with MOBI_IGEMM_WM=4
forcing the 256-pixel geometry on small problems.   python tools/check_pp.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOBI_IGEMM_WM"] = "4"
from mobi_amd import build, ops
build.build(verbose=False)
g = torch.Generator().manual_seed(1)
dt = torch.bfloat16
for (n, h, w, cin, cout, k, res, rowvec, blocks) in [
        (3, 16, 16, 64, 160, 1, False, False, 0), (3, 16, 16, 64, 160, 3, False, False, 0),
        (3, 16, 16, 256, 160, 3, True, False, 0), (4, 32, 32, 320, 320, 3, True, False, 0),
        (4, 32, 32, 320, 320, 3, True, False, 3), (4, 32, 32, 128, 128, 3, False, True, 5),
        (8, 8, 8, 128, 320, 3, True, True, 2), (2, 64, 64, 320, 640, 1, True, True, 7),
        (2, 64, 64, 320, 640, 1, False, False, 7), (2, 64, 64, 320, 640, 1, True, False, 0), (3, 16, 16, 192, 160, 1, False, False, 0),
        (3, 16, 16, 64, 160, 3, True, False, 0), (3, 16, 16, 128, 160, 3, False, False, 0),
        (1, 128, 128, 128, 128, 3, True, False, 0), (1, 64, 64, 192, 320, 3, False, True, 5), (2, 16, 512, 64, 128, 3, False, False, 0)]:
    if blocks: os.environ["MOBI_IGEMM_PERSIST_BLOCKS"] = str(blocks)
    else: os.environ.pop("MOBI_IGEMM_PERSIST_BLOCKS", None)
    x = torch.randn(n, h, w, cin, generator=g).to(dt)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).to(dt)
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, h, w, cout, generator=g).to(dt) if res else None
    rv = torch.randn(n, cout, generator=g) if rowvec else None
    pw = ops.pack_conv(wt.float(), None if rowvec else b, dt, "cuda")
    y = ops.igemm(x.cuda(), pw, residual=None if r is None else r.cuda(), rowvec=None if rv is None else rv.cuda(),
                  rowvec_has_bias=rowvec)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float(), None if rowvec else b, padding=k // 2).permute(0, 2, 3, 1)
    if rv is not None: ref = ref + rv[:, None, None, :]
    if r is not None: ref = ref + r.float()
    e = ((y.float().cpu() - ref).norm() / ref.norm()).item()
    bad = (~torch.isfinite(y.float())).sum().item()
    print(f"n={n} {h}x{w} {cin}->{cout} k{k} res={res} rowvec={rowvec} blocks={blocks}: rel={e:.2e} nonfinite={bad}", flush=True)
    d = (y.float().cpu() - ref).abs().reshape(-1, cout)
    badm = (d > 0.05 * ref.abs().max()) | ~torch.isfinite(d)
    if badm.any():
        idx = badm.nonzero()
        print("   bad elements:", len(idx), " pixels(m):", sorted(set(idx[:, 0].tolist()))[:24], " channels:", sorted(set(idx[:, 1].tolist()))[:40])
        y2 = ops.igemm(x.cuda(), pw, residual=None if r is None else r.cuda(), rowvec=None if rv is None else rv.cuda(), rowvec_has_bias=rowvec)
        d2 = (y2.float().cpu() - ref).abs().reshape(-1, cout)
        b2 = ((d2 > 0.05 * ref.abs().max()) | ~torch.isfinite(d2)).nonzero()
        print("   2nd run bad:", len(b2), " pixels:", sorted(set(b2[:, 0].tolist()))[:24])
