#!/bin/bash
# Interleaved A/B of whole configurations (several environment switches at once) on the denoising step, one box:
#   bash tools/ab_cfg.sh "MOBI_GROUPED_Q=0 MOBI_LN_FOLD=0" "MOBI_GROUPED_Q=1 MOBI_LN_FOLD=0" "MOBI_GROUPED_Q=1 MOBI_LN_FOLD=1"
flags="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-e2e --no-plms-line --no-fp16-line --no-config-lines"
for rep in $(seq 1 ${AB_REPS:-2}); do
  for wl in ${AB_WORKLOADS:-mobi_nusc_512 mobi_nusc_256}; do
    for cfg in "$@"; do
      ms=$(env $cfg python bench.py $flags --workload $wl 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
      echo "rep $rep $wl [$cfg]: $ms ms per step"
    done
  done
done
