#!/bin/bash
# GroupNorm / LayerNorm timings on the step's shapes (run before and after a change on the same box via git stash, or just after)
for c in 320 640 960; do python tools/kbench.py gn --c $c --hw 4096 --images 16 --iters 50; done
for c in 640 1280 1920; do python tools/kbench.py gn --c $c --hw 1024 --images 16 --iters 50; done
python tools/kbench.py gn --c 1280 --hw 256 --images 16 --iters 50
python tools/kbench.py gn --c 2560 --hw 256 --images 16 --iters 50
python tools/kbench.py ln --c 320 --hw 4096 --images 16 --iters 50
python tools/kbench.py ln --c 640 --hw 1024 --images 16 --iters 50
python tools/kbench.py ln --c 1280 --hw 256 --images 16 --iters 50
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from mobi_amd import ops
from tools.kbench import timeit
for n, t, c in ((16, 4096, 320), (16, 1024, 640), (16, 256, 1280)):
    x = torch.randn(n, t, c, device="cuda").bfloat16()
    a = torch.randn(n, 8, c, device="cuda") * 0.05; u = torch.randn(n, 8, c, device="cuda"); b = torch.randn(n, c, device="cuda")
    cc = torch.randn(n, 8, device="cuda"); asum = a.sum(-1).contiguous()
    us = timeit(lambda: ops.two_key_adapter(x, a, asum, cc, u, b, 1e-5), 50)
    print(f"two_key_adapter C={c} tokens={t} images={n}: {us:.1f} us  {4.0 * x.numel() / us / 1e3:.0f} GB/s (2 passes)")
PY
