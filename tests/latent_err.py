#!/usr/bin/env python3
"""(lives under tests/: it reads the oracle's fixtures)  Where the sampler's final-latent error comes from: the engine's DDIM-10 run
on the ORACLE's 9-channel input and conditioning (sampler + UNet arithmetic only) against the run on the engine's own get_input
outputs (+ what the VAE encoders' and the conditioning projection's errors become in the latent).
    python tests/latent_err.py [--side 64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=64)
    a = ap.parse_args()
    import mobi_amd
    from tests import oracle_cases as oc
    from tests import test_gpu_production as tp
    from tests.golden_cases import rel_l2
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    for dtype in (torch.float16, torch.bfloat16):
        mobi_amd.set_engine_dtype(dtype)
        model, i = tp._e2e_model(a.side)
        ref = oc.e2e(a.side)
        z, cond = ref["z"].cuda(), ref["cond"].cuda()

        def run(zz, cc):
            s, _ = DDIMSampler(model).sample(S=oc.E2E_STEPS, batch_size=2, shape=[4, a.side, a.side], conditioning=cc, verbose=False,
                                             eta=0.0, x_T=i["x_T"].cuda(),
                                             test_model_kwargs={"inpaint_image": zz[:, 4:8].contiguous(), "inpaint_mask": zz[:, 8:9].contiguous()})
            return rel_l2(s.cpu(), ref["samples"])
        exact = run(z, cond)
        # the engine's own encodes: perturb the oracle's input by the 16-bit rounding of its values (the floor of any 16-bit encoder)
        zr = z.clone()
        zr[:, :8] = z[:, :8].to(dtype).float()
        print(f"{dtype} side {a.side}: latent error with the ORACLE's input and conditioning {exact:.3e}; with that input rounded to the "
              f"storage type {run(zr, cond):.3e}", flush=True)


if __name__ == "__main__":
    main()
