#!/bin/bash
# How much of a ring-kernel launch's FETCH_SIZE is Infinity-Cache (MALL) hits, not HBM: the 3 x 3 convolution 320 -> 320 at 64 x 64 of
# the step (16 images: 42 MB in, 42 MB out, re-launched on the same buffers = resident in the 256-MiB MALL) against the same
# launch scaled PAST the MALL (112 images: 294 MB in + 294 MB out): counter bytes per algorithmic byte of both.  Separate --pmc passes.
#   bash tools/mall_share.sh            (outputs gpurun_out/mall/summary.txt)
set -u
OUT=gpurun_out/mall
mkdir -p $OUT
export TMPDIR=/tmp
for n in 16 112; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/n${n}_$c -- python3 tools/kbench.py conv --cin 320 --cout 320 --hw 64 --images $n --iters 4 \
      > $OUT/run_n${n}_$c.log 2> $OUT/err_n${n}_$c.log
  done
done
OUTDIR="$OUT" python3 - <<'PY'
import csv, glob, os
out = os.environ["OUTDIR"]
lines = ["# tools/mall_share.sh: 3 x 3 convolution 320 -> 320 at 64 x 64 (igemm_ring_kernel, 256 x 320 tiles), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes;",
         "# counter bytes = FETCH_SIZE x 2 (gfx950: 64 B tallied per 128-B request) + WRITE_SIZE, per launch, against the algorithmic bytes (input + weights + output)"]
for n in (16, 112):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, cnt = 0.0, 0
        for f in glob.glob(f"{out}/n{n}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "igemm_ring_kernel" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); cnt += 1
        vals[c] = (tot / max(cnt, 1), cnt)
    alg = (n * 4096 * 320 * 2 * 2 + 320 * 2880 * 2) / 1e6
    fetch_mb, write_mb = vals["FETCH_SIZE"][0] * 2 * 1024 / 1e6, vals["WRITE_SIZE"][0] * 1024 / 1e6
    lines.append(f"images {n:3d}: algorithmic {alg:7.1f} MB per launch | FETCH x2 {fetch_mb:7.1f} MB + WRITE {write_mb:7.1f} MB = {fetch_mb + write_mb:7.1f} MB "
                 f"= {(fetch_mb + write_mb) / alg:.2f} x algorithmic  ({vals['FETCH_SIZE'][1]} launches counted)")
with open(out + "/summary.txt", "w") as fh:
    fh.write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*counter_collection.csv" -delete
