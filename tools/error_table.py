#!/usr/bin/env python3
"""profiles/<tag>_error_table.txt from the records the GPU suite leaves under MOBI_RECORD_ERRORS:
    MOBI_RECORD_ERRORS=gpurun_out/errors.tsv python -m pytest tests -m gpu -q      (GPU box)
    python tools/error_table.py gpurun_out/errors.tsv profiles/r02_error_table.txt
Per test function: number of recorded comparisons and the LARGEST relative-L2 error per storage type (parametrisations
whose id carries dtype0 = fp16, dtype1 = bf16; tests without a dtype parameter run in fp16: marked fp16*)."""
import collections
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = collections.OrderedDict()
    named = collections.OrderedDict()
    for line in open(src):
        parts = line.rstrip("\n").split("\t")
        if len(parts) < 3:
            continue
        test, name, err = parts[0], parts[1], float(parts[2])
        if name != "rel_l2" and len(parts) > 3 and parts[3] != "nan":
            named[name] = (err, parts[3])
            continue
        m = re.match(r"(.*?)(\[(.*)\])?$", test)
        fn, pid = m.group(1), m.group(3) or ""
        col = "bf16" if ("dtype1" in pid or "bfloat16" in pid or "bfloat16" in name) else ("fp16" if ("dtype0" in pid or "float16" in name) else "fp16*")
        r = rows.setdefault(fn, {"n": 0, "fp16": None, "bf16": None, "fp16*": None})
        r["n"] += 1
        if err == err:
            r[col] = err if r[col] is None else max(r[col], err)
    with open(dst, "w") as f:
        f.write("Largest relative-L2 error of every parity assertion of `pytest tests -m gpu` on the MI355X (MOBI_RECORD_ERRORS),\n"
                "per test function and storage type, against the fp32 reference goldens / CPU oracle / fp32 torch of the same inputs.\n"
                "Asserted tolerances are <= 2x these values (fp16* = the test runs in fp16 storage only).\n\n")
        f.write(f"{'test':78s} {'cases':>6s} {'fp16':>10s} {'bf16':>10s}\n")
        for fn, r in rows.items():
            a = r["fp16"] if r["fp16"] is not None else r["fp16*"]
            star = "*" if r["fp16"] is None and r["fp16*"] is not None else ""
            fa = f"{a:.2e}{star}" if a is not None else "-"
            fb = f"{r['bf16']:.2e}" if r["bf16"] is not None else "-"
            f.write(f"{fn:78s} {r['n']:6d} {fa:>10s} {fb:>10s}\n")
        f.write("\nNamed quantities (measured, asserted bound):\n")
        for name, (err, tol) in named.items():
            f.write(f"  {name:70s} {err:.3e}   asserted < {tol}\n")
    print(f"wrote {dst}: {len(rows)} tests, {len(named)} named quantities")


if __name__ == "__main__":
    main()
