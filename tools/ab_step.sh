#!/bin/bash
# Interleaved A/B of one environment switch on the whole denoising step (bench.py's graph-replayed step, same box):
#   bash tools/ab_step.sh MOBI_GN_FUSED "" 0 1          (variable, then the values to compare; "" = unset)
set -e
var=$1; shift
flags="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line"
for rep in 1 2 3; do
  for wl in mobi_nusc_512 mobi_nusc_256; do
    for v in "$@"; do
      if [ -z "$v" ]; then unset $var; else export $var=$v; fi
      ms=$(python bench.py $flags --workload $wl 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
      echo "rep $rep $wl $var=${v:-unset}: $ms ms per step"
    done
  done
done
