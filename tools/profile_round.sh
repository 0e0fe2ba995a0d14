#!/bin/bash
# One GPU-box call: bench line, launch dump, rocprofv3 kernel stats and the two --pmc passes of the same command.
#   bash tools/profile_round.sh r01      (outputs under gpurun_out/<tag>/; copy the summaries into profiles/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --dump-launches $OUT/launches.tsv > $OUT/bench.json 2> $OUT/bench.err
# (the traced runs issue every launch from the host, --no-graph: the same kernels, one dispatch record each)
# (bf16 only: no fp16 sub-record in the traced runs, so every kernel row of the stats is the benched storage type)
TRACED="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e --no-graph --no-plms-line --no-fp16-line --no-config-lines"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $TRACED > $OUT/bench_trace.json 2> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-graph --no-plms-line --no-fp16-line --no-config-lines > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
done
python tools/pmc_summary.py $OUT/pmc_traffic.json FETCH_SIZE=$OUT/pmc_FETCH_SIZE WRITE_SIZE=$OUT/pmc_WRITE_SIZE > $OUT/pmc_summary.txt 2>&1
# the other reported lines of the same build: fp16 storage, classifier-free guidance, BASELINE config 2
python bench.py --dtype fp16 --no-cpu-baseline --no-e2e --no-plms-line --no-config-lines > $OUT/bench_fp16.json 2>> $OUT/bench.err
python bench.py --cfg-scale 5 --no-cpu-baseline --no-e2e --no-plms-line --no-config-lines > $OUT/bench_cfg5.json 2>> $OUT/bench.err
python bench.py --workload mobi_nusc_256 --steps 30 --no-cpu-baseline --dump-launches $OUT/launches256.tsv > $OUT/bench_256.json 2>> $OUT/bench.err
python bench.py --workload mobi_nusc_256 --steps 30 --no-cpu-baseline --no-e2e --no-graph --no-roofline --no-plms-line > $OUT/bench_256_nograph.json 2>> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the VAEs (A16 / A17): per-launch times of encode / decode of 8 images at 512 x 512, both autoencoders; the row-chain lab
python tools/vae_prof.py > $OUT/vae_prof.txt 2>&1
python tools/chain_lab.py > $OUT/chain_lab.txt 2>&1
python tools/chain_stamps.py > $OUT/chain_stamps.txt 2>&1
for f in 1 2 4 8 15; do python tools/chain_stamps.py -DMOBI_CHAIN_DBG=$f 2>&1 | grep launch >> $OUT/chain_stamps.txt; done
# the raw counter / trace CSVs are large: keep only the summaries for the merge back
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/bench.json
head -12 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt | head -40
