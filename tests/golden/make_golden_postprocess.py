#!/usr/bin/env python3
"""Generate tests/golden/postprocess.npz by running the REFERENCE's own `inverse_depth_normalization`
(ldm/data/utils.py:560-580) and the intensity expression of ddpm.py:1541-1543 on CPU (build container only).

    python tests/golden/make_golden_postprocess.py

Inputs are seeded (`oracle.weights.synth_input`) and include the branch boundaries (+-alpha, +-1) exactly.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import weights as W                      # noqa: E402
import make_golden as MG                             # noqa: E402  (stubs for the import-time packages)


def main():
    MG.import_reference()                              # stubs + the reference's `ldm` first on the path
    from ldm.data.utils import inverse_depth_normalization
    alpha = 0.75
    b, h, w = 3, 16, 24
    sample = torch.clamp(W.synth_input("post.sample", (b, 2, h, w)) * 0.8, -1.0, 1.0)
    edge = torch.tensor([-1.0, -alpha, alpha, 1.0, -0.75000006, 0.75000006, 0.0, 0.99999994])
    sample[0, 0, 0, :8] = edge
    sample[1, 1, 0, :4] = torch.tensor([1.0, -1.0, 0.99999994, 0.0])        # log(0) = -inf -> clamp
    min_d = torch.tensor([-0.6, -0.9, 0.1])
    max_d = torch.tensor([0.3, -0.2, 0.95])
    depth = sample[:, [0]].clone()
    for i in range(b):                                                        # ddpm.py:1533-1537
        depth[i] = inverse_depth_normalization(depth[i], min_d[i], max_d[i], alpha=alpha)
    inten = sample[:, [1]]
    inten = torch.clamp(-0.5 * torch.log(1 - (inten + 1) / 2) - 1, -1, 1)    # ddpm.py:1541
    MG.save("postprocess", sample=sample, min_d=min_d, max_d=max_d, alpha=np.float64(alpha), depth=depth, intensity=inten)


if __name__ == "__main__":
    main()
