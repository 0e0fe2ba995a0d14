#!/usr/bin/env python3
"""tests/golden/oracle_outputs.npz: the CPU oracle's outputs for the expensive GPU parity cases (tests/oracle_cases.py),
computed here once instead of on the GPU box at every `pytest -m gpu` run.   python tests/golden/make_oracle_outputs.py"""
import os
import sys
import time

os.environ["MOBI_ORACLE_LIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                          # noqa: E402
from tests import oracle_cases as oc                       # noqa: E402


def main():
    out = {}
    t0 = time.time()
    if "--add-vae512" in sys.argv:                      # keep what the file holds, add the 512 x 512 VAE cases only
        out = dict(np.load(oc.PATH))
        for lidar in (False, True):
            m, d = oc.vae512(lidar, live=True)
            tag = "lidar" if lidar else "camera"
            out[f"vae512_{tag}_moments"], out[f"vae512_{tag}_decode"] = m.numpy(), d.numpy().astype(np.float16)
            print(f"vae512 {tag}: {time.time() - t0:.0f} s", flush=True)
        np.savez_compressed(oc.PATH, **out)
        print(f"wrote {oc.PATH}: {sorted(out)} ({os.path.getsize(oc.PATH) / 1e6:.1f} MB)")
        return
    if "--add-e2e" in sys.argv:                         # keep what the file holds, add the end-to-end cases only
        out = dict(np.load(oc.PATH))
        for side in (32, 64):
            for k, v in oc.e2e(side, live=True).items():
                out[f"e2e{side}_{k}"] = v.numpy()
            print(f"e2e {side}: {time.time() - t0:.0f} s", flush=True)
        np.savez_compressed(oc.PATH, **out)
        print(f"wrote {oc.PATH}: {sorted(out)} ({os.path.getsize(oc.PATH) / 1e6:.1f} MB)")
        return
    if "--add-e2e-long" in sys.argv:                    # keep what the file holds, add the 50-step end-to-end cases
        out = dict(np.load(oc.PATH))
        cases = [(32, "ddim50"), (32, "plms50_cfg5"), (64, "ddim50")]
        for side, kind in cases:
            os.environ.pop("MOBI_ORACLE_LIVE", None)   # (z / cond of the stored 10-step case are read from the file)
            oc.LIVE = False
            for k, v in oc.e2e_long(side, kind, live=True).items():
                out[f"e2e{side}_{kind}_{k}"] = v.numpy().astype(np.float16 if k != "samples" else np.float32)
            print(f"e2e_long {side} {kind}: {time.time() - t0:.0f} s", flush=True)
            np.savez_compressed(oc.PATH, **out)
        print(f"wrote {oc.PATH}: {sorted(out)} ({os.path.getsize(oc.PATH) / 1e6:.1f} MB)")
        return
    out["prod_64_16"] = oc.prod_forward(64, 16, live=True).numpy()
    print(f"prod 64x64 x 16: {time.time() - t0:.0f} s", flush=True)
    out["prod_32_8"] = oc.prod_forward(32, 8, live=True).numpy()
    out["full_width16"] = oc.full_width16(live=True).numpy()
    for k, v in oc.trajectories10(live=True).items():
        out["traj10_" + k] = v.numpy()
    for lidar in (False, True):
        m, d = oc.vae512(lidar, live=True)
        tag = "lidar" if lidar else "camera"
        out[f"vae512_{tag}_moments"], out[f"vae512_{tag}_decode"] = m.numpy(), d.numpy().astype(np.float16)
    for side in (32, 64):
        for k, v in oc.e2e(side, live=True).items():
            out[f"e2e{side}_{k}"] = v.numpy()
    np.savez_compressed(oc.PATH, **out)
    print(f"wrote {oc.PATH}: {sorted(out)} ({os.path.getsize(oc.PATH) / 1e6:.1f} MB, {time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
