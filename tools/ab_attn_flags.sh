#!/bin/bash
# rebuild with each flag set in FLAGSETS (';'-separated) and time the attention launches of the 64x64 and 32x32 levels
set -u
IFS=';' read -ra SETS <<< "${FLAGSETS:- }"
for fs in "${SETS[@]}"; do
  MOBI_HIPCC_FLAGS="$fs" python -m mobi_amd.build --force > /tmp/ab_attn_build.log 2>&1 || { echo "build failed: $fs"; tail -5 /tmp/ab_attn_build.log; exit 1; }
  echo "== flags: $fs"
  python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 16 --v-rows --iters 10 2>&1 | grep attention
  python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 8 --v-rows --iters 10 2>&1 | grep attention
  python tools/kbench.py attn --heads 8 --dh 80 --t 1024 --images 16 --v-rows --iters 20 2>&1 | grep attention
  python tools/kbench.py attn --heads 8 --dh 160 --t 256 --images 16 --v-rows --iters 20 2>&1 | grep attention
done
python -m mobi_amd.build --force > /dev/null 2>&1
