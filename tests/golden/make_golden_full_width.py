#!/usr/bin/env python3
"""tests/golden/unet_full_width16.npz: the REFERENCE's own UNetModel at PRODUCTION width (model_channels 320, 1.04 B
parameters: configs/mobi_nusc_512.yaml:63-82) run on CPU on one camera / lidar pair at a 16 x 16 latent -- the case
`tests/oracle_cases.full_width16` holds from the oracle.  Pins the full-width graph (every block, every adapter, the 4x
channel multipliers) to the reference directly instead of through the reduced-width goldens.

Run in the build container only (`/root/reference` does not exist on the GPU box):

    python tests/golden/make_golden_full_width.py

Parameters are not stored: both sides regenerate them from `oracle.weights` (seed 5, keyed by state-dict name); the
inputs are `synth_input("uf.x")` / `synth_input("uf.c")`, t = 741.  Nothing of the reference is copied."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import weights as W                      # noqa: E402
import make_golden as mg                             # noqa: E402

KW = dict(image_size=64, in_channels=9, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
          num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
          context_dim=768, legacy=False, bbox_cond=True, use_camera=True, use_lidar=True)


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    t0 = time.time()
    R = mg.import_reference()
    net = R.om.UNetModel(**KW).eval()
    W.fill_module_(net, seed=5)
    print(f"reference UNetModel at full width: {sum(p.numel() for p in net.parameters()) / 1e9:.3f} B parameters, "
          f"filled in {time.time() - t0:.0f} s", flush=True)
    x = W.synth_input("uf.x", (2, 9, 16, 16))
    ctx = W.synth_input("uf.c", (2, 2, 768))
    t = torch.tensor([741, 741], dtype=torch.long)
    y = net(x, t, context=ctx)
    assert y.shape == (2, 4, 16, 16) and float(y.abs().max()) > 0
    mg.save("unet_full_width16", y=y, t=t, n_params=np.int64(sum(p.numel() for p in net.parameters())))
    print(f"done in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
