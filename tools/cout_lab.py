#!/usr/bin/env python3
"""Few-output-channel convolution: matrix-core form against the one-wave-per-pixel kernel (MOBI_COUT_MFMA=0) on the UNet's
and the VAE decoders' output convolutions (graph-timed, see tools/gn_lab.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gn_lab import timeit  # noqa: E402


def main():
    from mobi_amd import _lib, build, ops
    build.build(verbose=False)
    g = torch.Generator(device="cpu").manual_seed(0)
    for n, hw, cin, cout in ((16, 64, 320, 4), (8, 32, 320, 4), (8, 512, 128, 3), (8, 512, 128, 2)):
        xs = [torch.randn(n, hw, hw, cin, generator=g).cuda().to(torch.bfloat16) for _ in range(2)]
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        pw = ops.pack_conv(w, torch.zeros(cout), torch.bfloat16, "cuda")
        fns = [lambda x=x: ops.conv_small_cout(x, pw) for x in xs]
        res = {}
        for tag, env in (("matrix", None), ("per-pixel", "0")):
            if env is None:
                os.environ.pop("MOBI_COUT_MFMA", None)
            else:
                os.environ["MOBI_COUT_MFMA"] = env
            _lib.load().mobi_tuning_reload()
            res[tag] = min(timeit(fns, 8) for _ in range(2))
        os.environ.pop("MOBI_COUT_MFMA", None)
        _lib.load().mobi_tuning_reload()
        print(f"conv {cin}->{cout} 3x3 at {hw}x{hw} x{n}: " + " | ".join(f"{k} {v:8.1f} us" for k, v in res.items()))


if __name__ == "__main__":
    main()
