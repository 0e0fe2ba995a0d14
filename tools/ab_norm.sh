#!/bin/bash
# GroupNorm / LayerNorm timings on the step's shapes (run before and after a change on the same box via git stash, or just after)
for c in 320 640 960; do python tools/kbench.py gn --c $c --hw 4096 --images 16 --iters 50; done
for c in 640 1280 1920; do python tools/kbench.py gn --c $c --hw 1024 --images 16 --iters 50; done
python tools/kbench.py gn --c 1280 --hw 256 --images 16 --iters 50
python tools/kbench.py gn --c 2560 --hw 256 --images 16 --iters 50
python tools/kbench.py ln --c 320 --hw 4096 --images 16 --iters 50
python tools/kbench.py ln --c 640 --hw 1024 --images 16 --iters 50
python tools/kbench.py ln --c 1280 --hw 256 --images 16 --iters 50
