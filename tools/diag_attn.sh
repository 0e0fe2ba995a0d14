#!/bin/bash
# attention kernel with parts removed (wrong results, timing only): -DMOBI_ATTN_DBG bit 0 = no exp, 1 = tiles loaded once,
# 2 = no P.V MFMA, 3 = no K.Q^T MFMA
for v in ${VARIANTS:-0 1 2 4 8 12 13 15}; do
  MOBI_HIPCC_FLAGS="-DMOBI_ATTN_DBG=$v" python -m mobi_amd.build --force > /dev/null 2>&1 || { echo build $v failed; exit 1; }
  echo "== MOBI_ATTN_DBG=$v"
  python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 16 --v-rows --iters 10 2>&1 | grep attention
done
python -m mobi_amd.build --force > /dev/null 2>&1
