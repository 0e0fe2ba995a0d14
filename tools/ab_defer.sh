#!/bin/bash
# Split-K slabs summed by the consuming GroupNorm (ops.Deferred, mobi_split_source) against the reduce launches:
# parity tests first, then the interleaved whole-step A/B of both workloads on this box.
#   bash tools/ab_defer.sh > gpurun_out/ab_defer.txt
set -o pipefail
out=gpurun_out/ab_defer
mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_ops.py -x -q -m gpu \
  -k "sums_split_k or split_source or test_groupnorm or igemm_split_k" > $out/tests_ops.log 2>&1
rc=$?
tail -3 $out/tests_ops.log
[ $rc -ne 0 ] && { echo "op tests failed (rc=$rc)"; tail -40 $out/tests_ops.log; exit $rc; }
timeout -k 10 420 python -m pytest tests/test_gpu_production.py -x -q -m gpu -k "full_width_forward" > $out/tests_prod.log 2>&1
rc=$?
tail -3 $out/tests_prod.log
[ $rc -ne 0 ] && { echo "production tests failed (rc=$rc)"; tail -40 $out/tests_prod.log; exit $rc; }
bash tools/ab_cfg.sh "MOBI_DEFER_SPLIT=0" "MOBI_DEFER_SPLIT=1"
