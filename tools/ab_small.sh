#!/bin/bash
# Interleaved A/B of the small-problem igemm kernel on both steps (graph-replayed step of bench.py, one box), then the lab.
flags="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line"
for rep in 1 2 3; do
  for wl in mobi_nusc_256 mobi_nusc_512; do
    for v in 0 "" ; do
      if [ -z "$v" ]; then unset MOBI_IGEMM_SMALL; else export MOBI_IGEMM_SMALL=$v; fi
      ms=$(python bench.py $flags --workload $wl 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
      echo "rep $rep $wl MOBI_IGEMM_SMALL=${v:-unset}: $ms ms per step"
    done
  done
done
unset MOBI_IGEMM_SMALL
