#!/bin/bash
# Diagnosis of the direct-to-LDS main loop: rebuild the library with -DMOBI_DBG_SKIP=n (wrong results, timing only)
# and time three shapes per variant.  0 = shipped; 1 = no activation DMA; 2 = no weight DMA; 3 = no DMA at all;
# 4 = no MFMA; 7 = barrier / LDS-read skeleton only.
#   bash tools/diag_ingest.sh            (outputs under gpurun_out/diag_ingest/)
set -u
OUT=gpurun_out/diag_ingest
mkdir -p $OUT
for v in ${VARIANTS:-0 1 2 3 4 7}; do
  MOBI_HIPCC_FLAGS="-DMOBI_DBG_SKIP=$v" python -m mobi_amd.build --force > $OUT/build_$v.log 2>&1 || { echo "build $v failed"; tail -5 $OUT/build_$v.log; exit 1; }
  {
    echo "== MOBI_DBG_SKIP=$v"
    python tools/kbench.py conv --cin 320 --cout 320 --hw 64 --images 16 --iters 30
    python tools/kbench.py conv --cin 640 --cout 640 --hw 32 --images 16 --iters 30
    python tools/kbench.py linear --cin 320 --cout 320 --rows 65536 --residual --iters 50
    python tools/kbench.py linear --cin 1280 --cout 320 --rows 65536 --residual --iters 50
  } 2>&1 | tee -a $OUT/result.txt
done
python -m mobi_amd.build --force > $OUT/build_final.log 2>&1
