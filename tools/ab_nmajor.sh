# A/B of the igemm work-list order (MOBI_IGEMM_N_MAJOR=0: pixel tiles outermost always) on the weight-heavy shapes of a step
set -e
for rep in 1 2; do
  for x in "" 0; do
    for args in "conv --cin 1280 --cout 1280 --hw 16 --images 16" "conv --cin 2560 --cout 1280 --hw 16 --images 16" "conv --cin 1280 --cout 1280 --hw 8 --images 16" "conv --cin 2560 --cout 1280 --hw 8 --images 16" "conv --cin 640 --cout 640 --hw 32 --images 16" "conv --cin 1280 --cout 640 --hw 32 --images 16" "linear --cin 1280 --cout 1280 --rows 4096" "linear --cin 1280 --cout 10240 --rows 4096 --geglu"; do
      echo -n "rep $rep MOBI_IGEMM_N_MAJOR=${x:-unset}: "
      MOBI_IGEMM_N_MAJOR=$x python tools/kbench.py $args --iters 30 2>&1 | tail -1
    done
  done
done
