#!/bin/bash
# second pass of the slab-summing GroupNorm: op parity, in-graph breakdown, 8-byte-piece geometry on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "sums_split_k or split_source" > gpurun_out/tests_ops2.log 2>&1
rc=$?
tail -3 gpurun_out/tests_ops2.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/tests_ops2.log; exit $rc; }
bash tools/defer_breakdown.sh
AB_WORKLOADS=mobi_nusc_256 bash tools/ab_cfg.sh "MOBI_DEFER_SPLIT=0" "MOBI_DEFER_SPLIT=1" "MOBI_DEFER_SPLIT=1 MOBI_GN_SPLIT_PW4=1"
