#!/bin/bash
# rocprofv3 kernel stats of the two other measured paths of the round: the mobi_nusc_256 step (every launch host-issued) and one
# full-width training step.   bash tools/profile_extra.sh   (outputs under gpurun_out/extra/; copy the summaries into profiles/)
set -u
OUT=$PWD/gpurun_out/extra
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t256 -- python3 bench.py --workload mobi_nusc_256 --steps 5 --warmup 2 \
  --no-cpu-baseline --no-roofline --no-e2e --no-graph --no-plms-line --no-fp16-line > $OUT/bench256_trace.json 2> $OUT/t256.err
find $OUT/t256 -name "*kernel_stats.csv" -exec cp {} $OUT/nusc256_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ttrain -- python3 tools/train_bench.py --mc 320 --side 64 --n 16 --iters 2 \
  > $OUT/train_trace.log 2> $OUT/ttrain.err
find $OUT/ttrain -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
# HBM-side traffic of the mobi_nusc_256 step's kernels (separate --pmc passes, as the guide prescribes)
F256="--workload mobi_nusc_256 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-graph --no-plms-line --no-fp16-line"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc256_$c -- python3 bench.py $F256 > $OUT/bench256_pmc_$c.json 2> $OUT/pmc256_$c.err
done
python tools/pmc_summary.py $OUT/nusc256_pmc_traffic.json FETCH_SIZE=$OUT/pmc256_FETCH_SIZE WRITE_SIZE=$OUT/pmc256_WRITE_SIZE > $OUT/pmc256_summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
head -12 $OUT/nusc256_kernel_stats.csv | cut -c1-140
head -12 $OUT/train_kernel_stats.csv | cut -c1-140
tail -1 $OUT/train_trace.log
