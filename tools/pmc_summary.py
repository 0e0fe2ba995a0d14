#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter CSVs (one pass per counter) into profiles/<round>_pmc_traffic.json.

    python tools/pmc_summary.py OUT.json FETCH_SIZE=<dir> WRITE_SIZE=<dir> [...]

Per kernel family: launches, average counter value per launch; for the implicit-GEMM / attention kernels the HBM
bytes per launch = FETCH_SIZE x 2 (gfx950: a 128-byte request is tallied as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE,
both reported in KB.
"""
import csv
import glob
import json
import os
import sys

FAMILIES = ["igemm_pp_kernel", "igemm_ring64_kernel", "igemm_ring_kernel_128x160", "igemm_ring_kernel_128x320", "igemm_ring_kernel", "igemm_halo_kernel", "igemm_glds_kernel", "igemm_kernel",
            "igemm_splitk_reduce_kernel", "attention_rows_kernel", "attention_pipe_kernel", "attention_kernel", "ff_geglu_kernel",
            "gn_regs_kernel", "gn_stats_kernel", "gn_apply_kernel", "layernorm_kernel", "ctx_attention_kernel", "two_key_adapter",
            "row_chain_kernel", "small_gemm_kernel"]


def family(name):
    for f in FAMILIES:
        if f in name:
            if f == "igemm_ring_kernel":
                # one kernel name, three tile geometries (template arguments NW, MT; mangled `Li8ELi8E` or demangled `8, 8`): the
                # eight-wave 256 x 320 tiles keep the plain name (the bench line's dominant kernel), the others get their own
                if "Li4ELi4E" in name or ", 4, 4" in name:
                    return "igemm_ring_kernel_128x160"
                if "Li8ELi4E" in name or ", 8, 4" in name:
                    return "igemm_ring_kernel_128x320"
            return f
    return None


def fold(directory, counter):
    acc = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                fam = family(row["Kernel_Name"])
                if fam is None:
                    continue
                a = acc.setdefault(fam, {"launches": 0, "sum": 0.0})
                a["launches"] += 1
                a["sum"] += float(row["Counter_Value"])
    return {k: {"launches": v["launches"], "avg_per_launch": v["sum"] / v["launches"]} for k, v in acc.items()}


def main():
    out = sys.argv[1]
    raw = {}
    for spec in sys.argv[2:]:
        counter, directory = spec.split("=", 1)
        raw[counter] = fold(directory, counter)
    res = {"command": "rocprofv3 --pmc <counter> (one pass per counter) -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-e2e --no-cpu-baseline --no-roofline --no-graph --no-plms-line",
           # bench.py attaches `roofline.traffic` only to a run of this very configuration
           "config": {"workload": "mobi_nusc_512", "objects": 8, "dtype": "bf16", "cfg_scale": 1.0},
           "units": "FETCH_SIZE / WRITE_SIZE in KB per launch as reported; gfx950 correction (MI355X_MICROARCH.md, "
                    "HBM): FETCH_SIZE tallies 64 B per 128-B request of wide coalesced reads -> doubled",
           "raw": raw}
    res["note"] = ("igemm_ring_kernel = its eight-wave 256 x 320 tiles only; igemm_ring_kernel_128x160 / _128x320 = the four-wave / "
                   "eight-wave 128-pixel tiles of the same kernel template")
    for fam in FAMILIES:
        f, w = raw.get("FETCH_SIZE", {}).get(fam), raw.get("WRITE_SIZE", {}).get(fam)
        if f and w:
            res[fam] = {"launches": f["launches"], "fetch_kb_raw": f["avg_per_launch"], "write_kb": w["avg_per_launch"],
                        "hbm_bytes_per_launch_corrected": (2 * f["avg_per_launch"] + w["avg_per_launch"]) * 1024}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k in FAMILIES}, indent=1))


if __name__ == "__main__":
    main()
