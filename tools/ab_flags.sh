#!/bin/bash
# rebuild with each flag set in FLAGSETS (separated by ';') and time the conv / linear shapes: one box, same process order
set -u
OUT=gpurun_out/ab_flags
mkdir -p $OUT
IFS=';' read -ra SETS <<< "${FLAGSETS:- }"
for fs in "${SETS[@]}"; do
  MOBI_HIPCC_FLAGS="$fs" python -m mobi_amd.build --force > $OUT/build.log 2>&1 || { echo "build failed: $fs"; tail -5 $OUT/build.log; exit 1; }
  {
    echo "== flags: $fs"
    python tools/kbench.py conv --cin 320 --cout 320 --hw 64 --images 16 --iters 30
    python tools/kbench.py conv --cin 640 --cout 640 --hw 32 --images 16 --iters 30
    python tools/kbench.py conv --cin 1280 --cout 640 --hw 32 --images 16 --iters 30
    ${EXTRA_CMD:-true}
  } 2>&1 | grep -v amdgpu.ids | tee -a $OUT/result.txt
done
python -m mobi_amd.build --force > $OUT/build_final.log 2>&1
