#!/bin/bash
# In-graph cost of the slab-summing GroupNorm against reduce launch + plain GroupNorm, per kernel instantiation, both workloads:
#   bash tools/defer_breakdown.sh > gpurun_out/defer_breakdown.txt
set -u
OUT=$PWD/gpurun_out/db
mkdir -p $OUT
export TMPDIR=/tmp
F="--steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line --no-config-lines"
for wl in mobi_nusc_256 mobi_nusc_512; do
  for d in 0 1; do
    export MOBI_DEFER_SPLIT=$d
    rocprofv3 --kernel-trace --output-format csv -d $OUT/t_${wl}_$d -- python3 bench.py --workload $wl $F > $OUT/b_${wl}_$d.json 2> $OUT/t_${wl}_$d.err
    echo "== $wl MOBI_DEFER_SPLIT=$d: $(python -c "import json;print(json.loads(open('$OUT/b_${wl}_$d.json').read().strip().splitlines()[-1])['ms_per_step'])") ms per step"
    python tools/defer_breakdown.py $OUT/t_${wl}_$d
    find $OUT/t_${wl}_$d -name "*kernel_trace.csv" -delete
  done
done
