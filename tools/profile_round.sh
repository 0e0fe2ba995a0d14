#!/bin/bash
# One GPU-box call: bench line, launch dump, rocprofv3 kernel stats and the two --pmc passes of the same command.
#   bash tools/profile_round.sh r01      (outputs under gpurun_out/<tag>/; copy the summaries into profiles/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --dump-launches $OUT/launches.tsv > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e > $OUT/bench_trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
done
python tools/pmc_summary.py $OUT/pmc_traffic.json FETCH_SIZE=$OUT/pmc_FETCH_SIZE WRITE_SIZE=$OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw counter / trace CSVs are large: keep only the summaries for the merge back
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json
head -12 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt | head -40
