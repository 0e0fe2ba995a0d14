#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one engine source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kres.py igemm.hip [filter-substring] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = os.path.join(HERE, "mobi_amd", "csrc", args[0])
    flt = args[1] if len(args) > 1 else ""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:])
        sys.exit(1)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for name, d in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem and flt not in name:
            continue
        print(f"{dem[:110]:110s} vgpr={d.get('VGPRs', -1):3d} agpr={d.get('AGPRs', -1):3d} sgpr={d.get('TotalSGPRs', -1):3d} "
              f"spillV={d.get('VGPRs Spill', -1):3d} spillS={d.get('SGPRs Spill', -1):3d} "
              f"scratch={d.get('ScratchSize [bytes/lane]', -1):4d} lds={d.get('LDS Size [bytes/block]', -1)}")


if __name__ == "__main__":
    main()
