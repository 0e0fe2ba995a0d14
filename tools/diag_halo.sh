#!/bin/bash
# halo / ping-pong kernels with parts of the k loop removed (wrong results, timing only): -DMOBI_DBG_SKIP bit 0 = no
# activation DMA, bit 1 = no weight DMA, bit 2 = no MFMA
set -u
OUT=gpurun_out/diag_halo
mkdir -p $OUT
for v in ${VARIANTS:-0 1 2 3 7}; do
  MOBI_HIPCC_FLAGS="-DMOBI_DBG_SKIP=$v" python -m mobi_amd.build --force > $OUT/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 $OUT/build_$v.log; exit 1; }
  {
    echo "== MOBI_DBG_SKIP=$v"
    python tools/kbench.py conv --cin 640 --cout 640 --hw 32 --images 16 --iters 30
    MOBI_IGEMM_HALO=0 python tools/kbench.py conv --cin 640 --cout 640 --hw 32 --images 16 --iters 30
    python tools/kbench.py linear --cin 1280 --cout 320 --rows 65536 --residual --iters 50
  } 2>&1 | grep -v amdgpu.ids | tee -a $OUT/result.txt
done
python -m mobi_amd.build --force > $OUT/build_final.log 2>&1
