#!/bin/bash
# The round's closing GPU-box call: smoke, the whole GPU suite with its error records, then the profile set of the same build.
#   bash tools/final_round.sh r05      (outputs under gpurun_out/; tools/error_table.py + tools/collect_profiles.py copy them)
set -o pipefail
TAG=${1:-r05}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
rm -f gpurun_out/errors.tsv
MOBI_RECORD_ERRORS=gpurun_out/errors.tsv timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { echo "GPU suite failed (rc=$rc)"; tail -60 gpurun_out/gpu_tests.log; exit $rc; }
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
echo "profile rc=$?"
head -c 1500 gpurun_out/$TAG/bench.json
# the records tools/profile_more.sh refreshes that depend on the GroupNorm / split-K kernels: in-graph breakdown, the
# mobi_nusc_256 and training-step kernel stats (the Infinity-Cache and VAE-decoder records are other kernels': not re-run)
bash tools/graph_breakdown.sh > gpurun_out/$TAG/graph_breakdown.log 2>&1 && cp gpurun_out/gb/graph_breakdown.txt gpurun_out/$TAG/graph_breakdown.txt
bash tools/profile_extra.sh > gpurun_out/$TAG/profile_extra.log 2>&1
cp gpurun_out/extra/nusc256_kernel_stats.csv gpurun_out/extra/train_kernel_stats.csv gpurun_out/extra/nusc256_pmc_traffic.json gpurun_out/$TAG/ 2>/dev/null
echo; tail -1 gpurun_out/extra/train_trace.log
