#!/usr/bin/env python3
"""tests/golden/lr_schedules.npz: the REFERENCE's learning-rate schedule objects (`/root/reference/ldm/lr_scheduler.py`, pure
numpy: imported as it lies, nothing copied) evaluated on a fixed list of steps.   python tests/golden/make_golden_lr.py"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MOBI_REFERENCE", "/root/reference")

CASES = {
    # the schedule every MObI config sets (configs/mobi_nusc_512.yaml:54-61)
    "mobi_linear": ("LambdaLinearScheduler", dict(warm_up_steps=[200], cycle_lengths=[10000000000000], f_start=[1.e-6], f_max=[1.], f_min=[1.])),
    "linear_two_cycles": ("LambdaLinearScheduler", dict(warm_up_steps=[10, 5], cycle_lengths=[100, 50], f_start=[0.0, 0.1], f_max=[1.0, 0.5], f_min=[0.2, 0.05])),
    "cosine2_two_cycles": ("LambdaWarmUpCosineScheduler2", dict(warm_up_steps=[10, 5], cycle_lengths=[100, 50], f_start=[0.0, 0.1], f_max=[1.0, 0.5], f_min=[0.2, 0.05])),
    "cosine": ("LambdaWarmUpCosineScheduler", dict(warm_up_steps=20, lr_min=0.01, lr_max=1.0, lr_start=0.001, max_decay_steps=300)),
}
STEPS = {"mobi_linear": [0, 1, 2, 57, 199, 200, 201, 1000, 123456789], "linear_two_cycles": [0, 3, 9, 10, 11, 60, 99, 100, 101, 104, 105, 106, 130, 150],
         "cosine2_two_cycles": [0, 3, 9, 10, 11, 60, 99, 100, 101, 104, 105, 106, 130, 150], "cosine": [0, 5, 19, 20, 21, 150, 299, 300, 301, 5000]}


def main():
    spec = importlib.util.spec_from_file_location("ref_lr_scheduler", os.path.join(REF, "ldm", "lr_scheduler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, (cls, kw) in CASES.items():
        s = getattr(mod, cls)(**kw)
        out[name] = np.asarray([float(s(n)) for n in STEPS[name]], dtype=np.float64)
        out[name + "_steps"] = np.asarray(STEPS[name], dtype=np.int64)
    np.savez(os.path.join(HERE, "lr_schedules.npz"), **out)
    print("wrote lr_schedules.npz:", {k: v.shape for k, v in out.items() if not k.endswith("_steps")})


if __name__ == "__main__":
    main()
