#!/usr/bin/env python3
"""fp32-row linears of the time-embedding chain: matrix-core form against the vector-ALU kernel (MOBI_SKINNY_MFMA=0),
graph-timed (tools/gn_lab.py), with the largest difference between the two results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gn_lab import timeit  # noqa: E402


def main():
    from mobi_amd import _lib, build, ops
    from mobi_amd._lib import ACT_SILU
    build.build(verbose=False)
    g = torch.Generator(device="cpu").manual_seed(0)
    for m, k, n, pre in ((16, 320, 1280, 0), (16, 1280, 1280, 0), (16, 1280, 17600, ACT_SILU), (8, 1280, 17600, ACT_SILU), (2, 768, 1024, 0)):
        x = torch.randn(m, k, generator=g).cuda()
        ws = [(torch.randn(n, k, generator=g) * 0.03).cuda().to(torch.bfloat16) for _ in range(3)]
        b = torch.randn(n, generator=g).cuda()
        fns = [lambda w=w: ops.skinny_linear(x, w, b, pre_act=pre) for w in ws]
        res, outs = {}, {}
        for tag, env in (("matrix", None), ("vector", "0")):
            if env is None:
                os.environ.pop("MOBI_SKINNY_MFMA", None)
            else:
                os.environ["MOBI_SKINNY_MFMA"] = env
            _lib.load().mobi_tuning_reload()
            outs[tag] = fns[0]()
            res[tag] = min(timeit(fns, 12) for _ in range(2))
        os.environ.pop("MOBI_SKINNY_MFMA", None)
        _lib.load().mobi_tuning_reload()
        ref = torch.nn.functional.linear(torch.nn.functional.silu(x.double()) if pre else x.double(), ws[0].double(), b.double())
        err = {t: float((o.double() - ref).norm() / ref.norm()) for t, o in outs.items()}
        print(f"skinny m={m} k={k} n={n}: " + " | ".join(f"{t} {v:6.1f} us (rel err {err[t]:.1e})" for t, v in res.items()))


if __name__ == "__main__":
    main()
