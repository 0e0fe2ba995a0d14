#!/usr/bin/env python3
"""Measured relative-L2 error of the HIP engine against the CPU oracle (fp32 restatement of the reference graph),
per storage type, for the quantities BASELINE.json's north star names: one UNet forward, VAE encode / decode and
whole DDIM trajectories (10 and 50 steps, with and without classifier-free guidance).

    python tests/error_table.py [--out profiles/r01_error_table.txt]      (GPU box; about two minutes)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # (lives under tests/: it imports the oracle)
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import mobi_amd
    from oracle import sampler as osampler, unet as ounet, weights as W
    from tests.golden_cases import rel_l2
    from tests.test_gpu_models import _unet
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit("relative L2 error vs the fp32 CPU oracle (same seeded weights and inputs)")
    emit(f"{'quantity':58s} {'fp16':>10s} {'bf16':>10s}")

    def row(name, fn):
        vals = []
        for dt in (torch.float16, torch.bfloat16):
            mobi_amd.set_engine_dtype(dt)
            vals.append(fn())
        emit(f"{name:58s} {vals[0]:10.2e} {vals[1]:10.2e}")

    # ---- one UNet forward -------------------------------------------------------------------------------------
    for mc, side, batch in ((64, 16, 4), (64, 32, 2)):
        cfg = ounet.UNetConfig(model_channels=mc)
        sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 3)
        x = W.synth_input(f"u.x{mc}.{side}", (batch, 9, side, side))
        ctx = W.synth_input(f"u.c{mc}.{side}", (batch, 2, 768))
        t = torch.tensor([981, 1, 500, 21][:batch], dtype=torch.long)
        ref = ounet.unet_forward(sd, cfg, x, t, ctx)

        def f():
            net = _unet(cfg, side)
            net.load_state_dict(sd)
            return rel_l2(net.cuda()(x.cuda(), t.cuda(), context=ctx.cuda()).cpu(), ref)
        row(f"UNet forward, model_channels {mc}, latent {side}x{side}, batch {batch}", f)
    cfg = ounet.UNetConfig()
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 5)
    x = W.synth_input("uf.x", (2, 9, 16, 16))
    ctx = W.synth_input("uf.c", (2, 2, 768))
    t = torch.tensor([741, 741], dtype=torch.long)
    ref = ounet.unet_forward(sd, cfg, x, t, ctx)

    def f_full():
        net = _unet(cfg, 16)
        net.load_state_dict(sd)
        return rel_l2(net.cuda()(x.cuda(), t.cuda(), context=ctx.cuda()).cpu(), ref)
    row("UNet forward, mobi_nusc_512 width (320, 1.04 B par.), 16x16, batch 2", f_full)
    del sd, ref

    # ---- trajectories -----------------------------------------------------------------------------------------
    cfg = ounet.UNetConfig(model_channels=64)
    sd = W.synth_state_dict(ounet.unet_param_shapes(cfg), 9)
    b, side = 4, 16
    x_T = W.synth_input("smp.x_T", (b, 4, side, side))
    inp = W.synth_input("smp.inpaint", (b, 4, side, side))
    msk = (W.synth_input("smp.mask", (b, 1, side, side)) > 0).float()
    cond = W.synth_input("smp.cond", (b, 2, 768))
    uc = W.synth_input("smp.uc", (1, 2, 768)).repeat(b, 1, 1)
    eps = lambda xx, tt, cc: ounet.unet_forward(sd, cfg, xx, tt, cc)
    rest = torch.cat([inp, msk], 1)
    for S in (10, 50):
        sch = osampler.Schedule(S)
        for scale in (1.0, 5.0):
            t0 = time.time()
            ref, _ = osampler.ddim_sample(eps, sch, cond, x_T, rest, scale=scale, uncond=uc, log_every_t=1000)

            def f_traj():
                net = _unet(cfg, 16)
                net.load_state_dict(sd)
                net = net.cuda()

                class Model:
                    num_timesteps = 1000
                    device = torch.device("cuda")
                    betas = torch.from_numpy(sch.buffers["betas"]).cuda()
                    alphas_cumprod = torch.from_numpy(sch.buffers["alphas_cumprod"]).cuda()
                    alphas_cumprod_prev = torch.from_numpy(sch.buffers["alphas_cumprod_prev"]).cuda()

                    @staticmethod
                    def apply_model(xx, tt, cc):
                        return net(xx, tt, context=cc)
                s = DDIMSampler(Model())
                got, _ = s.sample(S=S, batch_size=b, shape=[4, side, side], conditioning=cond.cuda(), verbose=False,
                                  eta=0.0, x_T=x_T.cuda(), unconditional_guidance_scale=scale,
                                  unconditional_conditioning=uc.cuda(), log_every_t=1000,
                                  test_model_kwargs={"inpaint_image": inp.cuda(), "inpaint_mask": msk.cuda()})
                return rel_l2(got.cpu(), ref)
            row(f"DDIM-{S} final latent, cfg scale {scale:g} (mc 64, 16x16, batch 4)", f_traj)
    emit("tolerances asserted by tests/test_gpu_models.py: UNet / VAE fp16 5e-3, bf16 3e-2; 10-step trajectory fp16 2e-2, bf16 1e-1")
    if a.out:
        with open(a.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
