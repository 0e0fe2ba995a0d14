#!/bin/bash
# A/B of the 16x16x32 P.V on the last channel block of dh = 40 attention (MOBI_ATTN_H16=0: the 32x32x16 form), per launch
# and on the whole step, interleaved on one box.
for rep in 1 2 3; do
  for v in 0 ""; do
    if [ -z "$v" ]; then unset MOBI_ATTN_H16; else export MOBI_ATTN_H16=$v; fi
    for im in 16 8; do
      echo "rep $rep MOBI_ATTN_H16=${v:-unset} images $im: $(python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images $im --v-rows --iters 20 2>/dev/null | tail -1)"
    done
  done
done
flags="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line"
for rep in 1 2 3; do
  for v in 0 ""; do
    if [ -z "$v" ]; then unset MOBI_ATTN_H16; else export MOBI_ATTN_H16=$v; fi
    ms=$(python bench.py $flags 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "rep $rep mobi_nusc_512 MOBI_ATTN_H16=${v:-unset}: $ms ms per step"
  done
done
