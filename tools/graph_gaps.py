#!/usr/bin/env python3
"""Idle time between the kernels of a replayed step graph, from a rocprofv3 --kernel-trace CSV of bench.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python bench.py --steps 4 --warmup 1 ...
    python tools/graph_gaps.py gpurun_out/gaps
Takes the last `--launches` dispatches before the end of the trace window of the timed steps (the graph replays), sorts by
start time and reports busy time, gaps (start[i+1] - end[i], clipped at 0) and their distribution."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    # launches per step: given, or (`ms=<the traced bench line's ms_per_step>`) derived per run as dispatches x ms_per_step / span
    arg = sys.argv[2] if len(sys.argv) > 2 else "0"
    ms_per_step = float(arg[3:]) if arg.startswith("ms=") else 0.0
    n_per_step = 0 if ms_per_step else int(arg)
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    print(f"{len(rows)} dispatches in the trace")
    # the replayed steps: the longest run of dispatches whose gaps stay under 50 us
    runs, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] < 50_000:
            cur.append(b)
        else:
            runs.append(cur)
            cur = [b]
    runs.append(cur)
    runs.sort(key=len, reverse=True)
    for run in runs[:3]:
        busy = sum(e - s for s, e, _ in run)
        gaps = [max(0, b[0] - a[1]) for a, b in zip(run, run[1:])]
        span = run[-1][1] - run[0][0]
        if ms_per_step:
            n_per_step = max(1, round(len(run) * ms_per_step * 1e6 / span))
        gs = sorted(gaps)
        print(f"run of {len(run)} dispatches: span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, gaps {sum(gaps) / 1e6:.3f} ms "
              f"(median {gs[len(gs) // 2] / 1e3:.2f} us, p90 {gs[int(len(gs) * 0.9)] / 1e3:.2f} us, max {gs[-1] / 1e3:.1f} us)")
        if n_per_step:
            print(f"   per step of {n_per_step} launches: busy {busy / len(run) * n_per_step / 1e6:.3f} ms, gaps {sum(gaps) / len(run) * n_per_step / 1e6:.3f} ms")
        # in-graph time by kernel (template arguments stripped), per step
        if n_per_step:
            fam = {}
            for s0, e0, name in run:
                k = name.split("<")[0].split("(")[0]
                k = k[k.find("mobi"):] if "mobi" in k else k
                import re as _re
                k = _re.sub(r"^_ZN4mobi\d+", "", k)
                k = _re.sub(r"I(DF16_|DF16b).*", "", k)
                v = fam.setdefault(k[:48], [0, 0])
                v[0] += 1
                v[1] += e0 - s0
            steps = len(run) / n_per_step
            print(f"   in-graph time by kernel, per step ({steps:.2f} steps in the run):")
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
                print(f"      {k:48s} {v[0] / steps:7.1f} launches {v[1] / steps / 1e6:8.3f} ms  avg {v[1] / v[0] / 1e3:7.1f} us")
        # gaps by the kernel that FOLLOWS
        by = {}
        for (a, b), g in zip(zip(run, run[1:]), gaps):
            k = b[2].split("(")[0][:60]
            v = by.setdefault(k, [0, 0])
            v[0] += 1
            v[1] += g
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"      before {k:60s} n={v[0]:5d} avg gap {v[1] / v[0] / 1e3:6.2f} us")


if __name__ == "__main__":
    main()
