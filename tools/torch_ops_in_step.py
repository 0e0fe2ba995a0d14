#!/usr/bin/env python3
"""Which PyTorch (aten) kernels still run inside one denoising step, with shapes and the Python frame that issued
them: everything on the hot path should be an engine launch, torch is plumbing.   python tools/torch_ops_in_step.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    model = bench.build_model("mobi_nusc_512").cuda()
    sampler = DDIMSampler(model)
    sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    N, side = 16, 64
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).cuda()
    x, inpaint, cond = mk(N, 4, side, side), mk(N, 4, side, side), mk(N, 2, 768)
    mask = torch.ones(N, 1, side, side).cuda()
    kw = {"test_model_kwargs": {"inpaint_image": inpaint, "inpaint_mask": mask}}
    steps = list(reversed(sampler.ddim_timesteps.tolist()))

    def step(x, i):
        ts = torch.full((N,), int(steps[i]), device="cuda", dtype=torch.long)
        return sampler.p_sample_ddim(x, cond, ts, index=49 - i, **kw)[0]
    with torch.no_grad():
        for i in range(3):
            x = step(x, i)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                    record_shapes=True, with_stack=True) as prof:
            x = step(x, 3)
            torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True, group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=25,
                                                                                 max_src_column_width=110))


if __name__ == "__main__":
    main()
