#!/bin/bash
# The round's other measured records, one GPU-box call (outputs under gpurun_out/; tools/collect_profiles.py copies the summaries):
# per-kernel time inside the replayed step graph, the Infinity-Cache share of the ring kernel's counter traffic, the fp16 VAE decoders
# by precision option, the mobi_nusc_256 / training-step kernel stats.      bash tools/profile_more.sh r05
set -u
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/graph_breakdown.sh > $OUT/graph_breakdown.log 2>&1
cp gpurun_out/gb/graph_breakdown.txt $OUT/graph_breakdown.txt
bash tools/mall_share.sh > $OUT/mall_share.log 2>&1
cp gpurun_out/mall/summary.txt $OUT/mall_share.txt
( echo "# tools/vae_prof.py fp16: decode of 8 images at 512 x 512 by precision option (wall ms; MOBI_VAE_FP32_TRUNK / _FP32_STREAMS / _PRECISE_TAIL)"
  for cfg in "1 0 0" "1 1 0" "1 1 1"; do
    set -- $cfg
    echo "== trunk=$1 streams=$2 tail=$3"
    MOBI_VAE_FP32_TRUNK=$1 MOBI_VAE_FP32_STREAMS=$2 MOBI_VAE_PRECISE_TAIL=$3 python tools/vae_prof.py fp16 2>/dev/null | grep "^== "
  done ) > $OUT/vae_decode_fp16.txt
bash tools/profile_extra.sh > $OUT/profile_extra.log 2>&1
cp gpurun_out/extra/nusc256_kernel_stats.csv gpurun_out/extra/train_kernel_stats.csv gpurun_out/extra/nusc256_pmc_traffic.json $OUT/ 2>/dev/null
tail -1 gpurun_out/extra/train_trace.log > $OUT/train_step.txt 2>/dev/null
cat $OUT/mall_share.txt $OUT/vae_decode_fp16.txt
