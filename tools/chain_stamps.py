#!/usr/bin/env python3
"""Phase stamps of the row-chain kernel: builds csrc/chain.hip alone with -DMOBI_CHAIN_STAMPS (shader-clock stamps after the
staging and after every operation of one camera block and one lidar block), runs the post-attn1 program of the 64 x 64 level
through it and prints where a block's cycles go.     python tools/chain_stamps.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import mobi_amd  # noqa: E402
from mobi_amd import _lib, ops  # noqa: E402
from tools import _synth as W  # noqa: E402


def main():
    out = "/tmp/chain_stamps.so"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DMOBI_CHAIN_STAMPS",
           os.path.join(HERE, "mobi_amd", "csrc", "chain.hip"), "-o", out] + sys.argv[1:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    lab = C.CDLL(out)
    lab.mobi_row_chain.argtypes = [C.c_void_p, C.c_void_p]
    dt = torch.bfloat16
    mobi_amd.set_engine_dtype(dt)
    n, t, c = 16, 4096, 320
    mk = lambda name, *shape: W.synth_input(name, shape).to(dt).cuda()
    a, x = mk("st.a", n, t, c), mk("st.x", n, t, c)
    w = lambda name, **kw: ops.pack_chain_weight(W.synth_weight(name + ".weight", (c, c)), None, dt, "cuda", **kw)
    g, b = torch.ones(c), torch.zeros(c)
    cw = {k: w(k) for k in ("to_out", "k", "v")}
    cw["q"] = w("q", ln=(g, b), scale=0.2)
    rv = W.synth_input("st.rv", (n, c)).cuda().contiguous()
    tabs = ((W.synth_input("st.ta", (n, 8, c)) * 0.05).cuda(), None, W.synth_input("st.tc", (n, 8)).cuda(),
            W.synth_input("st.tu", (n, 8, c)).cuda(), W.synth_input("st.tb", (n, c)).cuda())
    tabs = (ops.chain_adapter_image(tabs[0], tabs[2], tabs[3], tabs[4], dt), 1e-5)
    new = lambda *s: torch.empty(s, device="cuda", dtype=dt)
    x1, q0, q1, kv = new(n, t, c), new(n // 2, t, c), new(n // 2, t, c), new(n // 2, t, 2 * c)
    head = lambda p: p.load(a, "s").load(x, "r").product(cw["to_out"], resid=True, to_s=True, bias=rv, bias_img_stride=c).adapter(dst=x1)
    p0 = head(ops.ChainProgram()).rowstats(1e-5).product(cw["q"], fold=True, dst=q0, dst_img_div=2)
    p1 = head(ops.ChainProgram()).rowstats(1e-5).product(cw["q"], fold=True, dst=q1, dst_img_div=2)
    p1.product(cw["k"], dst=kv[..., :c], dst_img_div=2).product(cw["v"], dst=kv[..., c:], dst_img_div=2)
    real = _lib.load().mobi_row_chain
    captured = {}

    def grab(pp, stream):
        captured["p"] = pp
        return real(pp, stream)
    lib = _lib.load()
    orig = lib.mobi_row_chain
    try:
        lib.mobi_row_chain = grab
        ops.row_chain([p0, p1], n, t, dt, adapter=tabs)
    finally:
        lib.mobi_row_chain = orig
    torch.cuda.synchronize()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lab.mobi_row_chain(captured["p"], st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lab.mobi_row_chain(captured["p"], st)
    e1.record()
    torch.cuda.synchronize()
    print(f"launch (post-attn1 program, [16, 4096, 320] bf16, flags {sys.argv[1:]}): {e0.elapsed_time(e1) * 50:.1f} us")
    buf = (C.c_ulonglong * 64)()
    assert lab.mobi_chain_debug_stamps(buf) == 0
    names = {0: ["(LOAD_S: hoisted)", "(LOAD_R: hoisted)", "PRODUCT to_out", "ADAPTER", "ROWSTATS", "PRODUCT q"],
             1: ["(LOAD_S: hoisted)", "(LOAD_R: hoisted)", "PRODUCT to_out", "ADAPTER", "ROWSTATS", "PRODUCT q", "PRODUCT k", "PRODUCT v"]}
    for img in (0, 1):
        s = list(buf[img * 32:(img + 1) * 32])
        print(f"--- {'camera' if img == 0 else 'lidar'} block: {s[2 + len(names[img]) - 1] - s[0]} cycles (100 MHz timer ticks x ?: see README) in all")
        print(f"   row loads + staging (first requests, vectors, tables)  {s[1] - s[0]:8d}")
        prev, pk = s[1], 0
        for i, nm in enumerate(names[img]):
            if nm.startswith("("):
                continue
            cur = s[2 + i]
            extra = ""
            if nm.startswith("PRODUCT"):
                extra = f"   (MFMA loop {s[20 + pk] - prev}, epilogue {s[16 + pk] - s[20 + pk]})"
                pk += 1
            print(f"   {nm:18s} {cur - prev:8d}{extra}")
            prev = cur


if __name__ == "__main__":
    main()
