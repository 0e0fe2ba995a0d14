#!/usr/bin/env python3
"""Single-kernel micro-benchmarks of the engine (for rocprofv3 --pmc / --kernel-trace runs and A/B timing).

    python tools/kbench.py attn   --heads 8 --dh 40 --t 4096 --images 16 [--dtype bf16] [--iters 20]
    python tools/kbench.py conv   --cin 640 --cout 640 --hw 64 --images 16 [--k 3] [--up] [--stride 1]
    python tools/kbench.py linear --cin 320 --cout 320 --rows 65536 [--residual] [--geglu]
    python tools/kbench.py gn     --c 320 --hw 4096 --images 16

Prints one line: kernel, shape, avg us, TFLOP/s or GB/s (algorithmic).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["attn", "conv", "linear", "gn", "ln", "tka", "ff"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--dh", type=int, default=40)
    ap.add_argument("--t", type=int, default=4096)
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--cin", type=int, default=320)
    ap.add_argument("--cin2", type=int, default=0)
    ap.add_argument("--cout", type=int, default=320)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--up", action="store_true")
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--geglu", action="store_true")
    ap.add_argument("--split", type=int, default=None)
    ap.add_argument("--chunk-major", action="store_true")
    ap.add_argument("--v-rows", action="store_true", help="attn: V row-major inside a stacked q|k|v tensor")
    ap.add_argument("--c", type=int, default=320)
    a = ap.parse_args()
    from mobi_amd import build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev).to(dt)
    if a.kind == "attn":
        c = a.heads * a.dh
        if a.v_rows:
            qkv = rn(a.images, a.t, 3 * c)
            q, k, vt = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
        else:
            q, k = rn(a.images, a.t, c), rn(a.images, a.t, c)
            vt = rn(a.images, c, a.t)
        us = timeit(lambda: ops.attention(q, k, vt, a.heads, a.dh ** -0.5, v_rows=a.v_rows), a.iters)
        fl = 4.0 * a.images * a.heads * a.t * a.t * a.dh
        print(f"attention heads={a.heads} dh={a.dh} T={a.t} images={a.images}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s")
    elif a.kind == "ff":
        # fused GEGLU feed-forward (mobi_ff_geglu) against the two launches it replaces
        c, hidden = a.cin, 4 * a.cin
        x, res = rn(a.rows, c), rn(a.rows, c)
        w1 = torch.randn(2 * hidden, c, generator=g) / c ** 0.5
        w2 = torch.randn(c, hidden, generator=g) / hidden ** 0.5
        b1, b2 = torch.zeros(2 * hidden), torch.zeros(c)
        pf = ops.pack_ff_geglu(w1, b1, w2, b2, dt, dev)
        pg, po = ops.pack_geglu(w1, b1, dt, dev), ops.pack_linear(w2, b2, dt, dev)
        x3, r3 = x.view(1, a.rows, c), res.view(1, a.rows, c)
        fused = lambda: ops.ff_geglu(x, pf, residual=res)
        two = lambda: ops.linear(ops.linear(x3, pg), po, residual=r3)
        best = {}
        for rep in range(3):
            for tag, fn in (("fused", fused), ("two launches", two)):
                best[tag] = min(best.get(tag, 1e30), timeit(fn, a.iters))
        fl = 2.0 * a.rows * c * 2 * hidden + 2.0 * a.rows * hidden * c
        d = float((fused().float() - two().view(a.rows, c).float()).norm() / two().float().norm())
        print(f"ff_geglu rows={a.rows} c={c}: " + " | ".join(f"{t} {v:.1f} us {fl / v / 1e6:.0f} TFLOP/s" for t, v in best.items()) +
              f" | rel diff {d:.2e}")
    elif a.kind == "conv":
        x = rn(a.images, a.hw, a.hw, a.cin)
        x2 = rn(a.images, a.hw, a.hw, a.cin2) if a.cin2 else None
        w = torch.randn(a.cout, a.cin + a.cin2, a.k, a.k, generator=g) / (a.k * (a.cin + a.cin2) ** 0.5)
        pw = ops.pack_conv(w, torch.zeros(a.cout), dt, dev, chunk_major=a.chunk_major)
        fn = lambda: ops.igemm(x, pw, x2=x2, stride=a.stride, upsample=a.up, split_k=a.split)
        us = timeit(fn, a.iters)
        y = fn()
        fl = 2.0 * y.numel() * (a.cin + a.cin2) * a.k * a.k
        print(f"conv {a.cin}+{a.cin2}->{a.cout} k{a.k} s{a.stride} up={a.up} {a.hw}x{a.hw} images={a.images}: "
              f"{us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s")
    elif a.kind == "linear":
        x = rn(1, a.rows, a.cin)
        res = rn(1, a.rows, a.cout) if a.residual else None
        if a.geglu:
            pw = ops.pack_geglu(torch.randn(2 * a.cout, a.cin, generator=g) / a.cin ** 0.5, torch.zeros(2 * a.cout), dt, dev)
        else:
            pw = ops.pack_linear(torch.randn(a.cout, a.cin, generator=g) / a.cin ** 0.5, torch.zeros(a.cout), dt, dev)
        us = timeit(lambda: ops.linear(x, pw, residual=res), a.iters)
        fl = 2.0 * a.rows * a.cin * pw.n_packed
        by = 2.0 * a.rows * (a.cin + a.cout * (2 if a.residual else 1))
        print(f"linear {a.cin}->{a.cout} rows={a.rows} geglu={a.geglu} residual={a.residual}: {us:.1f} us  "
              f"{fl / us / 1e6:.1f} TFLOP/s  {by / us / 1e3:.0f} GB/s")
    elif a.kind == "tka":
        # two-key adapter: matrix-core kernel vs the vector-ALU kernel (MOBI_TKA_MFMA=0), interleaved best of 3
        from mobi_amd import _lib
        x = rn(a.images, a.hw, a.c)
        f = lambda *s_: torch.randn(*s_, generator=g).to(dev)
        aa, u, b, cc = f(a.images, 8, a.c) * 0.05, f(a.images, 8, a.c), f(a.images, a.c), f(a.images, 8)
        fn = lambda: ops.two_key_adapter(x, aa, aa.sum(-1).contiguous(), cc, u, b, 1e-5)
        best, outs = {}, {}
        for rep in range(3):
            for tag, env in (("mfma", {}), ("valu", {"MOBI_TKA_MFMA": "0"})):
                os.environ.update(env)
                _lib.load().mobi_tuning_reload()
                outs[tag] = fn().float()
                best[tag] = min(best.get(tag, 1e30), timeit(fn, a.iters))
                for k_ in env:
                    os.environ.pop(k_, None)
        _lib.load().mobi_tuning_reload()
        by = 2.0 * x.numel() * 2
        d = float((outs["mfma"] - outs["valu"]).norm() / outs["valu"].norm())
        print(f"two_key_adapter C={a.c} tokens={a.hw} images={a.images}: " +
              " | ".join(f"{t} {v:.1f} us {by / v / 1e3:.0f} GB/s" for t, v in best.items()) + f" | rel diff {d:.2e}")
    elif a.kind == "ln":
        x = rn(a.images, a.hw, a.c)
        gam, bet = torch.ones(a.c, device=dev), torch.zeros(a.c, device=dev)
        us = timeit(lambda: ops.layernorm(x, gam, bet, 1e-5), a.iters)
        by = 2.0 * x.numel() * 2
        print(f"layernorm C={a.c} tokens={a.hw} images={a.images}: {us:.1f} us  {by / us / 1e3:.0f} GB/s (2 passes)")
    else:
        x = rn(a.images, 1, a.hw, a.c)
        gam, bet = torch.ones(a.c, device=dev), torch.zeros(a.c, device=dev)
        us = timeit(lambda: ops.groupnorm(x, gam, bet, 1e-5, True), a.iters)
        by = 2.0 * x.numel() * 3
        print(f"groupnorm+silu C={a.c} hw={a.hw} images={a.images}: {us:.1f} us  {by / us / 1e3:.0f} GB/s (3 passes)")


if __name__ == "__main__":
    main()
