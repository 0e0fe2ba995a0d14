#!/usr/bin/env python3
"""tests/golden/losses.npz: the loss side of the reference's training / validation step, produced by the REFERENCE's own
methods on CPU (build container only): `DDPM.register_schedule`'s `lvlb_weights` (ddpm.py:169-179) and
`LatentDiffusion.p_losses` (ddpm.py:1177-1217, `first_stage_key == 'inpaint'`: only the 4 latent channels are noised, the
other 5 pass through) with `DDPM.q_sample` / `get_loss`, called on a stand-in object whose `apply_model` returns a given
tensor (the UNet itself is pinned elsewhere).   python tests/golden/make_golden_losses.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402
from oracle import weights as W                           # noqa: E402


def main():
    R = mg.import_reference()
    D = R.ddpm

    class Stand(torch.nn.Module):
        pass
    for loss_type in ("l2", "l1"):
        m = Stand()
        m.v_posterior, m.parameterization = 0.0, "eps"
        D.DDPM.register_schedule(m, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120)
        m.first_stage_key, m.loss_type, m.learn_logvar = "inpaint", loss_type, False
        m.l_simple_weight, m.original_elbo_weight = 1.0, 0.25              # (the configs use 0: a non-zero weight covers the term)
        m.logvar = torch.full(fill_value=0.3, size=(m.num_timesteps,))
        m.q_sample = lambda x_start, t, noise=None, m=m: D.DDPM.q_sample(m, x_start, t, noise)
        m.get_loss = lambda pred, target, mean=True, m=m: D.DDPM.get_loss(m, pred, target, mean)
        seen = {}

        def apply_model(x_noisy, t, cond, seen=seen):
            seen["x_noisy"], seen["t"] = x_noisy.clone(), t.clone()
            return model_out
        m.apply_model = apply_model
        type(m).device = property(lambda self: torch.device("cpu"))
        m.eval()
        N = 6
        x_start = W.synth_input("loss.x", (N, 9, 8, 8))
        noise = W.synth_input("loss.noise", (N, 4, 8, 8))
        model_out = W.synth_input("loss.out", (N, 4, 8, 8)) * 0.7 + noise * 0.5
        t = torch.tensor([0, 1, 17, 500, 998, 999])
        cond = W.synth_input("loss.c", (N, 2, 768))
        loss, d = D.LatentDiffusion.p_losses(m, x_start, cond, t, noise=noise)
        out = {"x_start": x_start, "noise": noise, "model_out": model_out, "t": t, "x_noisy": seen["x_noisy"],
               "loss": loss, **{k.replace("/", "__"): v for k, v in d.items()}}
        if loss_type == "l2":
            out["lvlb_weights"] = m.lvlb_weights
            res = {k: np.asarray(v) for k, v in out.items()}
        else:
            res.update({"l1_" + k: np.asarray(v) for k, v in out.items() if k.startswith(("loss", "val__"))})
    path = os.path.join(HERE, "losses.npz")
    np.savez_compressed(path, **res)
    print(f"wrote {path}: {sorted(res)}")


if __name__ == "__main__":
    main()
