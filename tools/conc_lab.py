#!/usr/bin/env python3
"""Lab: does the chip run two captured denoising steps of independent half batches CONCURRENTLY (two streams), and is that
faster than one step of the whole batch?  (Every kernel pays a ramp and a tail; a second stream can fill them.)

    python tools/conc_lab.py [--workload mobi_nusc_512] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mobi_nusc_512")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    import bench
    import mobi_amd
    from mobi_amd import build, graph
    build.build(verbose=False)
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    mobi_amd.set_engine_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    wl = bench.WORKLOADS[args.workload]
    side, B = wl["latent"], wl["objects"]
    N = 2 * B
    dev = torch.device("cuda", 0)
    model = bench.build_model(args.workload).to(dev)
    g = torch.Generator(device="cpu").manual_seed(1234)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)

    def make(n):
        s = DDIMSampler(model)
        s.make_schedule(50, ddim_eta=0.0, verbose=False)
        s.refresh_weights_fingerprint()
        img, inp = mk(n, 4, side, side), mk(n, 4, side, side)
        mask = torch.ones(n, 1, side, side, device=dev)
        mask[:, :, side // 4: 3 * side // 4, side // 4: 3 * side // 4] = 0
        cond = mk(n, 2, 768)
        kw = {"test_model_kwargs": {"inpaint_image": inp, "inpaint_mask": mask}}
        with torch.no_grad():
            sg = graph.get(s, "ddim", img, cond, None, 1.0, kw)
            sg.run(img, 981, s._coef_table()[49], None)
        torch.cuda.synchronize()
        return s, sg, (img, cond, kw)

    def time_graphs(sgs, streams, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for sg, st in zip(sgs, streams):
                with torch.cuda.stream(st):
                    sg.graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    main_s = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    full = make(N)
    full2 = make(N)
    h1 = make(N // 2)
    h2 = make(N // 2)
    q = [make(N // 4) for _ in range(4)]
    s3, s4 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for rep in range(3):
        a = time_graphs([full[1]], [main_s], args.steps)
        c = time_graphs([h1[1]], [main_s], args.steps)
        b = time_graphs([h1[1], h2[1]], [s1, s2], args.steps)
        bs = time_graphs([h1[1], h2[1]], [main_s, main_s], args.steps)
        d = time_graphs([full[1], full2[1]], [s1, s2], args.steps)
        e = time_graphs([x[1] for x in q], [s1, s2, s3, s4], args.steps)
        print(f"rep {rep}: batch {N} one stream {a:.3f} ms | batch {N // 2} alone {c:.3f} | 2 x batch {N // 2} on two streams {b:.3f} "
              f"(same stream {bs:.3f}) | 2 x batch {N} on two streams {d:.3f} (= {d / 2:.3f} per batch-{N} step) | "
              f"4 x batch {N // 4} on four streams {e:.3f}", flush=True)


if __name__ == "__main__":
    main()
