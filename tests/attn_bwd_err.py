#!/usr/bin/env python3
"""(lives under tests/: it uses the tests' seeded inputs)  What a COMMON component of the keys / values (every token carrying the same
offset, as LayerNorm biases and smooth feature maps give them) does to the attention backward's 16-bit arithmetic: dq / dk / dv of
mobi_attention_bwd against fp64 autograd on the same rounded inputs, by size of the offset.
    python tests/attn_bwd_err.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mobi_amd import ops
    from oracle import weights as W
    n, heads, dh, t = 2, 8, 80, 1024
    c = heads * dh
    scale = dh ** -0.5
    for dtype in (torch.float16, torch.bfloat16):
        for koff, voff in ((0, 0), (0, 3), (0, 10), (3, 0), (10, 0), (3, 3), (10, 10)):
            mk = lambda name, off: (W.synth_input(name, (n, t, c)) + off * W.synth_input(name + ".off", (1, 1, c))).to(dtype)
            q, k, v = W.synth_input("ab.q", (n, t, c)).to(dtype), mk("ab.k", koff), mk("ab.v", voff)
            do = W.synth_input("ab.do", (n, t, c)).to(dtype)
            q64, k64, v64 = (x.double().clone().requires_grad_(True) for x in (q, k, v))
            sp = lambda x: x.reshape(n, -1, heads, dh).permute(0, 2, 1, 3)
            o = torch.einsum("bhij,bhjd->bhid", (torch.einsum("bhid,bhjd->bhij", sp(q64), sp(k64)) * scale).softmax(-1), sp(v64))
            o = o.permute(0, 2, 1, 3).reshape(n, t, c)
            o.backward(do.double())
            od = o.detach().to(dtype).cuda()
            dq, dk, dv = ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), od, do.cuda(), heads, scale)
            rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
            print(f"{str(dtype):15s} key offset {koff:2d} value offset {voff:2d}: dq {rel(dq, q64.grad):.2e}  dk {rel(dk, k64.grad):.2e}  "
                  f"dv {rel(dv, v64.grad):.2e}", flush=True)


if __name__ == "__main__":
    main()
