#!/usr/bin/env python3
"""Time one training step of the adapter parameters (mobi_amd/train.py: forward with a tape + the backward pass through the
whole UNet + AdamW on the 432 adapter tensors) -- informational: the backward kernels are a first slice, not tuned.

    python tools/train_bench.py [--mc 320] [--side 32] [--n 4] [--dtype bf16] [--iters 3]"""
import argparse
import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import mobi_amd  # noqa: E402
from mobi_amd import ops, train  # noqa: E402
from tools import _synth as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mc", type=int, default=320)
    ap.add_argument("--side", type=int, default=32)
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    mobi_amd.set_engine_dtype(dt)
    from mobi_amd.ldm.util import instantiate_from_config, load_config
    ucfg = load_config(os.path.join(HERE, "configs", "mobi_nusc_512.yaml"),
                       ["model.params.lidar_stage_config.params.ckpt_path=null"])["model"]["params"]["unet_config"]
    ucfg["params"]["model_channels"] = a.mc                   # (the UNet of configs/mobi_nusc_512.yaml; --mc narrows it for quick runs)
    ucfg["params"]["image_size"] = a.side
    net = instantiate_from_config(ucfg)
    W.fill_module_(net, seed=3)
    net = net.cuda()
    x = W.synth_input("tb.x", (a.n, 9, a.side, a.side)).cuda()
    ctx = W.synth_input("tb.c", (a.n, 2, 768)).cuda()
    noise = W.synth_input("tb.n", (a.n, 4, a.side, a.side)).cuda()
    t = torch.full((a.n,), 500, dtype=torch.long, device="cuda")
    opt = train.AdamW({k: p for k, p in net.named_parameters() if any(m in k for m in train.TRAINABLE_MARKERS)}, lr=1e-5)
    sink = []

    def step(profile=False):
        if profile:
            ops.set_profiler(sink)
        loss, grads = train.loss_and_gradients(net, x, t, ctx, noise, loss_scale=1.0 if dt == torch.bfloat16 else 2.0 ** round(math.log2(noise.numel() / 4)))
        grads.pop("__dcontext__", None)
        opt.step(grads)
        ops.set_profiler(None)
        return loss
    with torch.no_grad():
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            loss = step()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / a.iters
        torch.cuda.reset_peak_memory_stats()
        step()
        torch.cuda.synchronize()
    n_par = sum(p.numel() for p in opt.params.values())
    print(f"training step, UNet model_channels {a.mc}, latent {a.side}x{a.side}, UNet batch {a.n}, {a.dtype}: {dt_s * 1e3:.1f} ms "
          f"(forward with tape + backward + AdamW on {len(opt.params)} tensors / {n_par / 1e6:.1f} M parameters), loss {float(loss):.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
