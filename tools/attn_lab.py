#!/usr/bin/env python3
"""Attention kernel lab: build attention.hip alone with a list of -D variants (seconds each), time every variant on the same
box in one process (interleaved, best of N rounds) and check it against fp32 torch.

    python tools/attn_lab.py [--dh 40 --t 4096 --images 16 --heads 8 --dtype bf16] \
        base="" ahead="-DMOBI_ATTN_RVAR=1" noexp="-DMOBI_ATTN_RDBG=1" ...

A variant is name="hipcc flags"; a name starting with '!' is a timing-only ablation (wrong results by design, no check).
Environment variables of the library (MOBI_ATTN_NW, ...) can be set per variant as name="ENV:MOBI_ATTN_NW=4 -D...".
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
from mobi_amd import _lib  # noqa: E402  (struct layouts only)


def build(tag, flags):
    out = f"/tmp/attn_lab_{tag}.so"
    csrc = os.path.join(HERE, "mobi_amd", "csrc")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-Wno-unused-result",
           os.path.join(csrc, "attention.hip"), os.path.join(csrc, "tuning.hip"), "-o", out] + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-4000:])
        raise SystemExit(f"build of variant {tag} failed")
    return C.CDLL(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dh", type=int, default=40)
    ap.add_argument("--t", type=int, default=4096)
    ap.add_argument("--tk", type=int, default=0)
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--qscale", type=float, default=1.0)
    ap.add_argument("--stamp", default="", help="hipcc flags of ONE variant built with -DMOBI_ATTN_STAMP: prints the "
                    "s_memtime stamps of waves 0 and NW/2 of block (1,1,1), steps 16..23 (attention_pipe_kernel)")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    tk = a.tk or a.t
    c = a.heads * a.dh
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = (torch.randn(a.images, max(a.t, tk), 3 * c, generator=g)).to("cuda").to(dt)
    qkv[..., :c] *= a.qscale
    q, k, v = qkv[:, :a.t, :c], qkv[:, :tk, c:2 * c], qkv[:, :tk, 2 * c:]
    out = torch.empty(a.images, a.t, c, device="cuda", dtype=dt)
    scale = a.dh ** -0.5
    # reference on a subset of images (fp32 torch on the device)
    nref = min(a.images, 2)
    sp = lambda t_: t_[:nref].float().reshape(nref, -1, a.heads, a.dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * scale
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(nref, a.t, c)
    del sim

    p = _lib.AttentionParams()
    p.q, p.q_img_stride, p.q_row_stride = q.data_ptr(), q.stride(0), q.stride(1)
    p.k, p.k_img_stride, p.k_row_stride = k.data_ptr(), k.stride(0), k.stride(1)
    p.vt, p.vt_img_stride, p.vt_row_stride = v.data_ptr(), v.stride(0), v.stride(1)
    p.v_layout = 1
    p.out, p.out_img_stride, p.out_row_stride = out.data_ptr(), out.stride(0), out.stride(1)
    p.images, p.heads, p.dh, p.tq, p.tk, p.scale = a.images, a.heads, a.dh, a.t, tk, scale
    p.dtype = _lib.MOBI_BF16 if dt == torch.bfloat16 else _lib.MOBI_F16
    stream = torch.cuda.current_stream().cuda_stream

    if a.stamp is not None and a.stamp != "":
        dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
        os.environ["MOBI_ATTN_STAMP_PTR"] = hex(dbg.data_ptr())
        lib = build("stamp", ["-DMOBI_ATTN_STAMP"] + [f for f in a.stamp.split() if f != "-"])
        lib.mobi_attention.argtypes = [C.c_void_p, C.c_void_p]
        for _ in range(3):
            assert lib.mobi_attention(C.byref(p), stream) == 0
        torch.cuda.synchronize()
        d = dbg.cpu().view(2, 8, 8)
        names = ["top", "staged", "reads issued", "MFMA gaps", "fix", "rotate", "barrier"]
        print("# ns between stamps (s_memrealtime, 10 ns ticks), steps 16..23; columns:", ", ".join(f"{names[i]}->{names[i + 1]}" for i in range(6)),
              ", barrier->next top, whole step")
        for w, tag in ((0, "wave 0 (early)"), (1, "wave NW/2 (late)")):
            for st in range(7):
                row = [int(d[w, st, i + 1] - d[w, st, i]) for i in range(6)]
                row.append(int(d[w, st + 1, 0] - d[w, st, 6]))
                row.append(int(d[w, st + 1, 0] - d[w, st, 0]))
                print(f"{tag:18s} step {16 + st}: " + " ".join(f"{x:6d}" for x in row))
        print("# late - early offsets of the step tops:", [int(d[1, st, 0] - d[0, st, 0]) for st in range(8)])
        return
    variants = []
    for spec in a.variants or ['base=']:
        name, _, flags = spec.partition("=")
        env = {}
        toks = flags.split()
        if toks and toks[0].startswith("ENV:"):
            for kv in toks[0][4:].split(","):
                kk, _, vv = kv.partition("=")
                env[kk] = vv
            toks = toks[1:]
        lib = build(name.lstrip("!"), toks)
        lib.mobi_attention.argtypes = [C.c_void_p, C.c_void_p]
        variants.append((name, lib, env))

    def run(lib, env):
        for kk, vv in env.items():
            os.environ[kk] = vv
        lib.mobi_tuning_reload()
        rc = lib.mobi_attention(C.byref(p), stream)
        assert rc == 0, rc
        for _ in range(2):
            lib.mobi_attention(C.byref(p), stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            lib.mobi_attention(C.byref(p), stream)
        e1.record()
        torch.cuda.synchronize()
        for kk in env:
            os.environ.pop(kk, None)
        return e0.elapsed_time(e1) * 1e3 / a.iters

    best, err = {}, {}
    for name, lib, env in variants:
        out.zero_()
        run(lib, env)
        y = out[:nref].float()
        err[name] = float((y - ref).norm() / ref.norm()) if not name.startswith("!") else float("nan")
    for _ in range(a.rounds):
        for name, lib, env in variants:
            best[name] = min(best.get(name, 1e30), run(lib, env))
    fl = 4.0 * a.images * a.heads * a.t * tk * a.dh
    print(f"# attention [{a.images}, {a.t} x {tk}, {a.heads} x {a.dh}] {a.dtype}, V row-major; best of {a.rounds} x {a.iters}")
    for name, _, _ in variants:
        print(f"{name:24s} {best[name]:8.1f} us  {fl / best[name] / 1e6:7.1f} TFLOP/s   rel err {err[name]:.2e}")


if __name__ == "__main__":
    main()
