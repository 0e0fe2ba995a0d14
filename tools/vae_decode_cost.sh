#!/bin/bash
# What the fp16 parity configuration's decoder options cost: tools/vae_prof.py fp16 (decode of 8 images at 512 x 512, wall ms) by
# MOBI_VAE_FP32_TRUNK / _FP32_STREAMS / _PRECISE_TAIL / MOBI_VAE_PRECISE (0: 16-bit operands; 1: activations split; 2: weights too = the default)
echo "# tools/vae_decode_cost.sh: tools/vae_prof.py fp16, decode of 8 images at 512 x 512 by precision option (wall ms)"
for cfg in "1 0 0 0" "1 1 0 0" "1 1 1 0" "1 1 1 1" "1 1 1 2"; do
  set -- $cfg
  echo "== trunk=$1 streams=$2 tail=$3 precise=$4"
  MOBI_VAE_FP32_TRUNK=$1 MOBI_VAE_FP32_STREAMS=$2 MOBI_VAE_PRECISE_TAIL=$3 MOBI_VAE_PRECISE=$4 python tools/vae_prof.py fp16 2>/dev/null | grep "^== "
done
