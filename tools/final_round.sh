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
