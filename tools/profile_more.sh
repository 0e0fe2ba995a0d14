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
bash tools/vae_decode_cost.sh > $OUT/vae_decode_fp16.txt
bash tools/profile_extra.sh > $OUT/profile_extra.log 2>&1
cp gpurun_out/extra/nusc256_kernel_stats.csv gpurun_out/extra/train_kernel_stats.csv gpurun_out/extra/nusc256_pmc_traffic.json $OUT/ 2>/dev/null
tail -1 gpurun_out/extra/train_trace.log > $OUT/train_step.txt 2>/dev/null
cat $OUT/mall_share.txt $OUT/vae_decode_fp16.txt
