#!/usr/bin/env python3
"""(lives under tests/: it reads the oracle's fixtures)  Where the end-to-end pixel error comes from: the engine's decoders on the ORACLE's final latent (decoder arithmetic only)
against the same decoders on the engine's own latent (+ what the sampler's latent error becomes in pixel space).
    python tests/decoder_err.py [--side 64]"""
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
    for dtype in (torch.float16, torch.bfloat16):
        mobi_amd.set_engine_dtype(dtype)
        model, i = tp._e2e_model(a.side)
        ref = oc.e2e(a.side)
        # the oracle's own latent through the engine's decoders
        z = ref["z"]
        z_lidar = torch.cat([z[1:2, :4]], 0)                       # (square range view: the aligned latent is the latent)
        h_cam, h_lid = model.decode_sample(ref["samples"].cuda(), z_lidar.cuda())
        for trunk, streams, tail, every in (("0", "0", "0", "0"), ("1", "0", "0", "0"), ("1", "1", "0", "0"), ("1", "1", "1", "0"),
                                            ("1", "1", "1", "1"), ("1", "1", "1", "2")):
            from mobi_amd.ldm.modules.diffusionmodules import model as M
            M._TRUNK_ENV, M._STREAMS_ENV, M._TAIL_ENV, M._PRECISE_ENV = trunk, streams, tail, every
            image = model.decode_first_stage(h_cam.contiguous(), clamp=(-1., 1.))
            rng = model.decode_first_stage(h_lid.contiguous(), module_name="lidar_stage_model", clamp=(-1., 1.))
            print(f"{dtype} side {a.side} trunk={trunk} streams={streams} tail={tail} every_conv_level={every}: decoders on the ORACLE's latent: camera "
                  f"{rel_l2(image.float().cpu(), ref['image']):.3e}  range {rel_l2(rng.float().cpu(), ref['range']):.3e}", flush=True)


if __name__ == "__main__":
    main()
