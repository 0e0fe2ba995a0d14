#!/bin/bash
# A/B of the ping-pong schedule (MOBI_IGEMM_PP=0 -> lockstep direct-to-LDS kernel) on the step's main shapes, one box.
set -u
OUT=gpurun_out/ab_pp
mkdir -p $OUT
run() {
  python tools/kbench.py conv --cin 320 --cout 320 --hw 64 --images 16 --iters 30
  python tools/kbench.py conv --cin 640 --cout 640 --hw 32 --images 16 --iters 30
  python tools/kbench.py conv --cin 1280 --cout 640 --hw 32 --images 16 --iters 30
  python tools/kbench.py linear --cin 320 --cout 320 --rows 65536 --residual --iters 50
  python tools/kbench.py linear --cin 1280 --cout 320 --rows 65536 --residual --iters 50
  python tools/kbench.py linear --cin 320 --cout 1280 --rows 65536 --geglu --iters 50
  python tools/kbench.py linear --cin 640 --cout 640 --rows 16384 --residual --iters 50
}
for pp in 1 0; do
  echo "== MOBI_IGEMM_PP=$pp"
  MOBI_IGEMM_PP=$pp run 2>&1 | grep -v amdgpu.ids
done | tee $OUT/result.txt
