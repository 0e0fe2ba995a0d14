# A/B of the XCD-aware workgroup map of the attention kernels (MOBI_ATTN_XCD=0: hardware order), shapes of a mobi_nusc_512 step
set -e
for rep in 1 2 3; do
  for x in "" 0; do
    for args in "--dh 40 --t 4096 --images 16" "--dh 40 --t 4096 --images 8" "--dh 80 --t 1024 --images 16" "--dh 160 --t 256 --images 16"; do
      echo -n "rep $rep MOBI_ATTN_XCD=${x:-unset}: "
      MOBI_ATTN_XCD=$x python tools/kbench.py attn --v-rows $args --iters 30 2>&1 | grep attention
    done
  done
done
