"""tests/golden/oracle_outputs.npz (the CPU oracle's outputs the `-m gpu` suite reads instead of recomputing them on the
GPU box) against the oracle run live, on a sample sized for the CPU suite: one camera/lidar pair of the mobi_nusc_256
production batch (full-width UNet, 32 x 32), the full-width 16 x 16 pair, and the DDIM-10 trajectory without guidance."""
import os

import numpy as np
import torch

from tests import oracle_cases as oc


def _close(a, b):
    # the same fp32 graph; thread count changes the order of a few reductions
    return float((a - b).norm() / b.norm()) < 1e-5


def test_file_holds_every_case():
    f = dict(np.load(oc.PATH))
    assert f["prod_64_16"].shape == (16, 4, 64, 64) and f["prod_32_8"].shape == (8, 4, 32, 32)
    assert f["full_width16"].shape == (2, 4, 16, 16) and int(f["traj10_n_pred_x0"]) > 0
    for tag, ch in (("camera", 3), ("lidar", 2)):
        assert f[f"vae512_{tag}_moments"].shape == (1, 8, 64, 64) and f[f"vae512_{tag}_decode"].shape == (1, ch, 512, 512)
        assert np.isfinite(f[f"vae512_{tag}_moments"]).all() and np.isfinite(f[f"vae512_{tag}_decode"].astype(np.float32)).all()
    for k in ("ddim_1.0", "ddim_5.0", "plms_1.0", "plms_5.0", "mask_eta1"):
        assert f["traj10_" + k].shape == (4, 4, 16, 16) and np.isfinite(f["traj10_" + k]).all()
    # the 50-step end-to-end cases (tests/oracle_cases.py e2e_long: ~45 minutes of CPU, generated once): present, finite, and
    # really other trajectories than the 10-step case they share their inputs with
    for side, kind in ((32, "ddim50"), (32, "plms50_cfg5"), (64, "ddim50")):
        R = 8 * side
        smp = f[f"e2e{side}_{kind}_samples"]
        assert smp.shape == (2, 4, side, side) and smp.dtype == np.float32 and np.isfinite(smp).all()
        assert f[f"e2e{side}_{kind}_image"].shape == (1, 3, R, R) and f[f"e2e{side}_{kind}_range"].shape == (1, 2, R, R)
        assert np.abs(f[f"e2e{side}_{kind}_image"].astype(np.float32)).max() <= 1.0
        assert float(np.abs(smp - f[f"e2e{side}_samples"]).max()) > 1e-3


def test_sample_recomputed_live():
    f = dict(np.load(oc.PATH))
    pair = oc.prod_forward(32, 8, pairs=[2], live=True)                    # elements 4, 5 (t = 501); ~2 min: the 1.04 B
    assert _close(pair, torch.from_numpy(f["prod_32_8"][4:6]))             # synthetic parameters dominate
    if os.environ.get("MOBI_ORACLE_FULL") == "1":                          # (another parameter set: opt-in)
        assert _close(oc.full_width16(live=True), torch.from_numpy(f["full_width16"]))
    tr = oc.trajectories10(live=True, only=("ddim_1.0",))
    assert _close(tr["ddim_1.0"], torch.from_numpy(f["traj10_ddim_1.0"])) and int(tr["n_pred_x0"]) == int(f["traj10_n_pred_x0"])


def test_full_width_oracle_output_against_the_reference():
    """The oracle's full-width case (1.04 B parameters, one camera / lidar pair at 16 x 16) against the REFERENCE's own
    UNetModel at production width (tests/golden/unet_full_width16.npz, made by tests/golden/make_golden_full_width.py from
    /root/reference): the production-width graph is pinned to the reference directly, not only through the reduced-width
    goldens.  Measured 1.6e-6 (two fp32 graphs, different summation orders)."""
    ref = np.load(os.path.join(os.path.dirname(oc.PATH), "unet_full_width16.npz"))
    assert int(ref["n_params"]) == 1039929604 and ref["y"].shape == (2, 4, 16, 16)
    a, b = torch.from_numpy(dict(np.load(oc.PATH))["full_width16"]).double(), torch.from_numpy(ref["y"]).double()
    assert float((a - b).norm() / b.norm()) < 1e-5
