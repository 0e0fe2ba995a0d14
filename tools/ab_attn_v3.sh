# A/B of the attention kernels on the shapes of a mobi_nusc_512 step (tools/kbench.py attn --v-rows)
set -e
for nw in "" 4 8; do
  echo "== MOBI_ATTN_NW=$nw"
  for args in "--dh 40 --t 4096 --images 16" "--dh 40 --t 4096 --images 8" "--dh 80 --t 1024 --images 16" "--dh 160 --t 256 --images 16"; do
    MOBI_ATTN_NW=$nw python tools/kbench.py attn --v-rows $args --iters 30 2>&1 | grep attention
  done
done
