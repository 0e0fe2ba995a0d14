#!/usr/bin/env python3
"""Fused feed-forward lab: build csrc/ff.hip alone with a list of -D variants, time them interleaved on one box.

    python tools/ff_lab.py base= "!nogelu=-DMOBI_FF_DBG=1" "!nodma=-DMOBI_FF_DBG=2" ...

A name starting with '!' is a timing-only ablation (wrong results, no check)."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from mobi_amd import _lib, ops  # noqa: E402


def build(tag, flags):
    out = f"/tmp/ff_lab_{tag}.so"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-Wno-unused-result",
           os.path.join(HERE, "mobi_amd", "csrc", "ff.hip"), "-o", out] + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-4000:])
        raise SystemExit(f"build of variant {tag} failed")
    return C.CDLL(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    c, hidden = 320, 1280
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(a.rows, c, generator=g).cuda().to(dt)
    res = torch.randn(a.rows, c, generator=g).cuda().to(dt)
    w1 = torch.randn(2 * hidden, c, generator=g) / c ** 0.5
    w2 = torch.randn(c, hidden, generator=g) / hidden ** 0.5
    b1, b2 = torch.randn(2 * hidden, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1
    pf = ops.pack_ff_geglu(w1, b1, w2, b2, dt, "cuda")
    ref = ops.ff_geglu(x, pf, residual=res).float()
    out = torch.empty_like(x)
    p = _lib.FfGegluParams()
    p.x, p.rows, p.c, p.hidden, p.w_packed, p.b2 = x.data_ptr(), a.rows, c, hidden, pf.buf.data_ptr(), pf.b2.data_ptr()
    p.residual, p.out, p.dtype = res.data_ptr(), out.data_ptr(), _lib.MOBI_BF16 if dt == torch.bfloat16 else _lib.MOBI_F16
    stream = torch.cuda.current_stream().cuda_stream
    variants = []
    for spec in a.variants or ["base="]:
        name, _, flags = spec.partition("=")
        lib = build(name.lstrip("!"), flags.split())
        lib.mobi_ff_geglu.argtypes = [C.c_void_p, C.c_void_p]
        variants.append((name, lib))

    def run(lib):
        for _ in range(3):
            assert lib.mobi_ff_geglu(C.byref(p), stream) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            lib.mobi_ff_geglu(C.byref(p), stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    best, err = {}, {}
    for name, lib in variants:
        out.zero_()
        run(lib)
        err[name] = float((out.float() - ref).norm() / ref.norm())
    for _ in range(3):
        for name, lib in variants:
            best[name] = min(best.get(name, 1e30), run(lib))
    fl = 2.0 * a.rows * c * 2 * hidden + 2.0 * a.rows * hidden * c
    print(f"# ff_geglu rows={a.rows} c={c} hidden={hidden} {a.dtype}; best of 3 x {a.iters}; rel diff against the shipped library")
    for name, _ in variants:
        print(f"{name:20s} {best[name]:8.1f} us  {fl / best[name] / 1e6:7.1f} TFLOP/s   rel diff {err[name]:.2e}")


if __name__ == "__main__":
    main()
