#!/bin/bash
# per-kernel time INSIDE the replayed step graph of both workloads (rocprofv3 --kernel-trace of bench.py + tools/graph_gaps.py)
set -u
OUT=$PWD/gpurun_out/gb
mkdir -p $OUT
export TMPDIR=/tmp
F="--steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line --no-config-lines"
rocprofv3 --kernel-trace --output-format csv -d $OUT/g512 -- python3 bench.py $F > $OUT/b512.json 2> $OUT/g512.err
python tools/graph_gaps.py $OUT/g512 ms=$(python -c "import json;print(json.loads(open('$OUT/b512.json').read().strip().splitlines()[-1])['ms_per_step'])") > $OUT/breakdown512.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/g256 -- python3 bench.py --workload mobi_nusc_256 $F > $OUT/b256.json 2> $OUT/g256.err
python tools/graph_gaps.py $OUT/g256 ms=$(python -c "import json;print(json.loads(open('$OUT/b256.json').read().strip().splitlines()[-1])['ms_per_step'])") > $OUT/breakdown256.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
( echo "# per-kernel time INSIDE the replayed step graph (rocprofv3 --kernel-trace of bench.py, tools/graph_gaps.py): no host issue time, no event brackets"
  echo "== mobi_nusc_512"; head -34 $OUT/breakdown512.txt; echo; echo "== mobi_nusc_256"; head -34 $OUT/breakdown256.txt ) > $OUT/graph_breakdown.txt
head -36 $OUT/graph_breakdown.txt | cut -c1-150
