#!/usr/bin/env python3
"""Row-chain lab: one BasicTransformerBlock at C = 320 (the 64 x 64 level of mobi_nusc_512: 16 images x 4096 tokens) with the
launches between its attention kernels chained (csrc/chain.hip) against the one-by-one sequence, interleaved on one box;
per-launch times of both sequences from the launch profiler (events around every launch of one eager pass).

    python tools/chain_lab.py [--n 16] [--side 64] [--dtype bf16] [--iters 10]"""
import argparse
import os
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import mobi_amd  # noqa: E402
from mobi_amd import ops  # noqa: E402
from mobi_amd.ldm.modules import attention as A  # noqa: E402
from tools import _synth as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--side", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    mobi_amd.set_engine_dtype(dt)
    blk = A.BasicTransformerBlock(320, 8, 40, context_dim=768, bbox_cond=True, multimodal=True)
    W.fill_module_(blk, seed=41)
    blk = blk.cuda()
    t = a.side * a.side
    xs = [(W.synth_input(f"lab.x{i}", (a.n, t, 320))).to(dt).cuda() for i in range(4)]     # rotate: cold inputs
    ctx = W.synth_input("lab.ctx", (a.n, 2, 768)).cuda()
    A.ROW_CHAIN_MIN_ROWS = 1

    def run(chained, iters):
        A.ROW_CHAIN = chained
        for i in range(2):
            blk(xs[i % 4], context=ctx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            blk(xs[i % 4], context=ctx)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    def launches(chained):
        A.ROW_CHAIN = chained
        sink = []
        ops.set_profiler(sink)
        blk(xs[0], context=ctx)
        torch.cuda.synchronize()
        ops.set_profiler(None)
        return [(k, fl, e0.elapsed_time(e1) * 1e3, nb, tag) for k, fl, e0, e1, nb, tag in sink]

    y1 = blk(xs[0], context=ctx)
    A.ROW_CHAIN = False
    y0 = blk(xs[0], context=ctx)
    print(f"chained vs one-by-one: rel-L2 {float((y1.float() - y0.float()).norm() / y0.float().norm()):.3e}")
    best = {}
    for _ in range(3):
        for ch in (False, True):
            best[ch] = min(best.get(ch, 1e30), run(ch, a.iters))
    print(f"block [{a.n}, {t}, 320] {a.dtype}: one-by-one {best[False]:.1f} us, chained {best[True]:.1f} us per block pass (host-issued)")
    for ch in (False, True):
        ls = launches(ch)
        print(f"--- {'chained' if ch else 'one-by-one'}: {len(ls)} launches, {sum(l[2] for l in ls):.1f} us in events")
        for k, fl, us, nb, tag in ls:
            print(f"   {k:16s} {us:8.1f} us  {fl / us / 1e6 if us else 0:7.1f} TF/s  {nb / us / 1e6 if us else 0:6.2f} TB/s  {tag}")


if __name__ == "__main__":
    main()
