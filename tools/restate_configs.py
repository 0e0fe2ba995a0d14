#!/usr/bin/env python3
"""Writes configs/*.yaml: the hyper-parameter VALUES of the reference's config files (parsed, re-emitted in this repo's
own formatting; `${}` interpolations kept as they are), so that BASELINE.json's five configs can be named on the GPU
box, where /root/reference does not exist.  tests/test_dropin_cpu.py checks value equality against the reference files
whenever they are present.      python tools/restate_configs.py [/root/reference/configs]"""
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["mobi_nusc-mini_256", "mobi_nusc-mini_512", "mobi_nusc_256", "mobi_nusc_512", "mobi_nusc_all-classes_256",
         "mobi_nusc_all-classes_512", "range_autoencoder", "pbe"]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/configs"
    for name in NAMES:
        with open(os.path.join(src, name + ".yaml")) as f:
            cfg = yaml.safe_load(f)
        head = (f"# {name}: hyper-parameter values of the reference's configs/{name}.yaml (tools/restate_configs.py).\n"
                "# `target: ldm.*` strings resolve to the engine's classes (top-level `ldm` aliases `mobi_amd.ldm`).\n")
        with open(os.path.join(ROOT, "configs", name + ".yaml"), "w") as f:
            f.write(head)
            yaml.safe_dump(cfg, f, sort_keys=False, default_flow_style=None, width=110)
        print("wrote", name)


if __name__ == "__main__":
    main()
