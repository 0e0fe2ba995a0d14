#!/usr/bin/env python3
"""In-graph time of the GroupNorm / split-K-reduce kernels by instantiation, from a rocprofv3 --kernel-trace CSV of bench.py's
replayed step graph (tools/defer_breakdown.sh): what the slab-summing GroupNorm costs against reduce launch + plain GroupNorm.
    python tools/defer_breakdown.py <trace dir> <steps in the run>"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    runs, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] < 50_000:
            cur.append(b)
        else:
            runs.append(cur)
            cur = [b]
    runs.append(cur)
    run = max(runs, key=len)
    span = (run[-1][1] - run[0][0]) / 1e6
    agg = collections.OrderedDict()
    for s, e, name in run:
        if "gn_" in name or "splitk_reduce" in name:
            m = re.search(r"gn_regs_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
            if m:
                key = f"gn_regs<{m.group(2)},{m.group(3)},{m.group(4)},{'SLAB' if m.group(5) == '1' else 'plain'}>"
            else:
                m = re.search(r"gn_regs_kernel<[^,]*, (\d+), (\d+), (\d+), (true|false)>", name)
                key = (f"gn_regs<{m.group(1)},{m.group(2)},{m.group(3)},{'SLAB' if m.group(4) == 'true' else 'plain'}>" if m
                       else ("splitk_reduce" if "splitk_reduce" in name else name[:60]))
            a = agg.setdefault(key, [0, 0])
            a[0] += 1
            a[1] += e - s
    print(f"run of {len(run)} dispatches, span {span:.3f} ms")
    tot = 0.0
    for k, (n, ns) in sorted(agg.items()):
        print(f"  {k:36s} {n:5d} launches  {ns / 1e6:8.3f} ms  avg {ns / n / 1e3:6.2f} us")
        tot += ns
    print(f"  GroupNorm + reduce kernels together: {tot / 1e6:.3f} ms of the run ({100 * tot / 1e6 / span:.1f} %)")


if __name__ == "__main__":
    main()
