#!/usr/bin/env python3
"""Two-key adapter kernels side by side on the shapes of one denoising step: token rows in registers (default), the
LDS-tile matrix-core kernel (MOBI_TKA_MFMA=1: C <= 640), the vector-ALU kernel (MOBI_TKA_MFMA=0).  Interleaved best of 3
replays of a captured graph of the launches; the inputs rotate over distinct tensors (cold reads).

    python tools/tka_lab.py [--images 16] [--dtype bf16] [--iters 24] [--rows N]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gn_lab import timeit  # noqa: E402

SHAPES_512 = [(4096, 320, 5), (1024, 640, 6), (256, 1280, 5), (64, 1280, 1)]   # tokens per image, C, launches per step
SHAPES_256 = [(1024, 320, 5), (256, 640, 6), (64, 1280, 5), (16, 1280, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--rows", default="", help="comma list of MOBI_TKA_ROWS values to sweep for the register kernel")
    ap.add_argument("--valu-rows", default="", help="the same for the vector-ALU kernel")
    ap.add_argument("--nusc256", action="store_true", help="the shapes of mobi_nusc_256 (use with --images 8)")
    a = ap.parse_args()
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    forms = [("registers", {}), ("lds-tile", {"MOBI_TKA_MFMA": "1"}), ("valu", {"MOBI_TKA_MFMA": "0"})]
    forms += [(f"registers/{r}", {"MOBI_TKA_ROWS": r}) for r in a.rows.split(",") if r]
    forms += [(f"valu/{r}", {"MOBI_TKA_ROWS": r, "MOBI_TKA_MFMA": "0"}) for r in a.valu_rows.split(",") if r]
    total = {t: 0.0 for t, _ in forms}
    print(f"two-key adapter, {a.images} images, {a.dtype}; us per launch (GB/s at 2 B read + 2 B written per element)")
    for t, c, n in (SHAPES_256 if a.nusc256 else SHAPES_512):
        mb = a.images * t * c * 2 / 1e6
        copies = max(2, min(12, int(600 / mb)))
        xs = [(torch.randn(a.images, t, c, generator=g) * 1.5 + 0.3).to("cuda").to(dt) for _ in range(2)]
        xs += [xs[i % 2].clone() for i in range(copies - 2)]
        f = lambda *s_: torch.randn(*s_, generator=g).cuda()
        aa, u, b, cc = f(a.images, 8, c) * 0.05, f(a.images, 8, c), f(a.images, c), f(a.images, 8)
        asum = aa.sum(-1).contiguous()
        fns = [lambda x=x: ops.two_key_adapter(x, aa, asum, cc, u, b, 1e-5) for x in xs]
        x32 = xs[0].float()
        mean = x32.mean(-1, keepdim=True)
        rstd = (x32.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        z = rstd * (torch.einsum("ntc,nhc->nth", x32, aa) - mean * asum[:, None, :]) + cc[:, None, :]
        ref = x32 + b[:, None, :] + torch.einsum("nth,nhc->ntc", torch.sigmoid(z), u)
        best, err = {}, {}
        for rep in range(3):
            for tag, env in forms:
                os.environ.update(env)
                _lib.load().mobi_tuning_reload()
                if rep == 0:
                    err[tag] = float((fns[0]().float() - ref).norm() / ref.norm())
                best[tag] = min(best.get(tag, 1e30), timeit(fns, a.iters))
                for k in env:
                    os.environ.pop(k, None)
        _lib.load().mobi_tuning_reload()
        for tag in total:
            total[tag] += best[tag] * n
        print(f"tokens={t:5d} C={c:5d} x{n} {mb:6.1f} MB: " +
              " | ".join(f"{tag} {best[tag]:6.1f} us {2 * mb / best[tag] * 1e3:5.0f} GB/s err {err[tag]:.1e}" for tag, _ in forms))
    print("per step (17 launches): " + " | ".join(f"{t} {v / 1e3:.3f} ms" for t, v in total.items()))


if __name__ == "__main__":
    main()
