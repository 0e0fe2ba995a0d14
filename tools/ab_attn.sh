#!/bin/bash
# A/B of the attention schedules on the 64x64-level launch, one box: MOBI_ATTN_SP=1 software-pipelined | 0 lockstep 8-wave kernel
for rep in 1 2; do
for pp in 1 0; do   # 1 = software-pipelined kernel, 0 = default
  echo "== MOBI_ATTN_SP=$pp"
  MOBI_ATTN_SP=$pp python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 16 --v-rows --iters 10 2>&1 | grep attention
  MOBI_ATTN_SP=$pp python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 8 --v-rows --iters 10 2>&1 | grep attention
done
done
