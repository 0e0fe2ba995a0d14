#!/usr/bin/env python3
"""Benchmark of the MObI sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mobi_nusc_512|mobi_nusc_256]
                    [--objects B] [--dtype bf16|fp16] [--cfg-scale S] [--e2e] [--no-cpu-baseline]

A "step" is one DDIM denoising step of the whole per-GPU batch: UNet forward on
2*B interleaved camera/lidar elements (4*B with classifier-free guidance) plus the fp32
latent update.  Synthetic inputs already resident in HBM, random-init weights of the
mobi_nusc_512 architecture (1.04 B parameters); there is no checkpoint / dataset offline.

Prints ONE JSON line (rank 0): metric = UNet element-forwards per second over all GPUs
(= denoising steps/s x UNet batch elements), with `steps_per_s`, `roofline` (dominant
kernel = the implicit-GEMM conv/linear kernel, algorithmic FLOPs / measured launch time
against the 2.5 PFLOP/s dense bf16 MFMA peak) and `cpu_baseline` (the CPU oracle --
the reference graph in PyTorch CPU fp32 -- timed on this host on a bounded sample).
Multi-GPU: `--gpus N` with no WORLD_SIZE in the environment spawns N ranks itself
(`python -m torch.distributed.run`, one process per GPU, RCCL) BEFORE this process touches the
GPU, and exits with the child's code; under an external torchrun the ranks run directly.  Objects
are sharded, no collective inside the denoising loop; the end-to-end pass adds VAE encode/decode
and the all-gather of decoded images.  The step is issued as ONE HIP graph launch
(mobi_amd/graph.py); `host_ms_per_step` is the time the host needs to issue a step,
`gpu_ms_per_step` the HIP-event time of the same region.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # BASELINE.json configs[2] / configs[1]
    # gflop_skippable: work of the reference graph the engine removes by exact algebra and therefore does not
    # claim: attn2's one-key softmax (25.6 / 6.4 GF) + the to_out/connector folds (4*T*C^2 per block: 25.6 / 6.4 GF)
    # + the two-key bbox adapter's to_q and folded output projection (another 4*T*C^2 per block: 26.8 / 6.7 GF)
    "mobi_nusc_512": dict(latent=64, objects=8, gflop_per_element=1021.9, gflop_skippable=78.0),
    "mobi_nusc_256": dict(latent=32, objects=4, gflop_per_element=209.7, gflop_skippable=19.5),
}
PMC_TRAFFIC = "r05_pmc_traffic.json"   # written by tools/pmc_summary.py from the --pmc passes of this round's build
PEAK_TFLOPS = 2500.0          # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


PARITY_FILES = ("gpurun_out/parity_last.json", "profiles/r05_parity.json")
# untimed steps in front of every SUB-record's timed steps: building a sampler and capturing its graph leaves the GPU idle, and an idle
# GPU drops its clocks (profiles/r05_e2e_idle_gap.txt): ~0.15 s of steps bring them back (x 4 at the 32 x 32 latent, whose step is 6 ms)
SUB_WARMUP_STEPS = 8
NORTH_STAR_REL_L2 = 1e-3


def lib_sha16():
    """sha256 of the engine library this process loads (first 16 hex digits): what a parity record is tied to."""
    import hashlib
    from mobi_amd import build
    with open(build.LIB, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def parity_record(side, dtype):
    """End-to-end parity at production width as the GPU suite last MEASURED it (tests/test_gpu_production.py::
    test_end_to_end_pixel_space writes gpurun_out/parity_last.json; tools/collect_profiles.py copies it to profiles/ with the
    commit) -- never re-measured here (the oracle run behind it is minutes of CPU).  Every record carries the hash of the library
    it was measured with: `matches_this_build` says whether that is the library of THIS run.  `meets` = per storage type, whether
    the latent AND both decoded pictures are within the north star's 1e-3 rel-L2 of the CPU oracle."""
    for rel_path in PARITY_FILES:
        try:
            with open(os.path.join(ROOT, rel_path)) as f:
                pj = json.load(f)
            case = "mobi_nusc_512" if side == 64 else "mobi_nusc-mini_256"
            key = next(k for k in pj if k.startswith(case))
            rec = pj[key]

            def one(d):
                r = rec[d]
                return {"dtype": d, "latent_rel_l2": r["latent_rel_l2"], "pixel_rel_l2_camera": r["pixel_rel_l2_camera"],
                        "pixel_rel_l2_range": r["pixel_rel_l2_range"]}

            other = "fp16" if dtype == "bf16" else "bf16"
            sha, ssha = pj.get("lib_sha16"), pj.get("src_sha16")
            from mobi_amd import build as _build
            same_bin, same_src = bool(sha) and sha == lib_sha16(), bool(ssha) and ssha == _build.sources_sha16()
            # (a rebuild in another directory need not reproduce the binary's hash: the hash over the kernel sources + flags is
            #  the identity that survives it)
            parity = {**one(dtype), "other_storage_type": one(other), "case": key, "source": rel_path,
                      "measured_with_lib_sha16": sha, "measured_with_src_sha16": ssha, "measured_at_commit": pj.get("commit"),
                      "matches_this_binary": same_bin, "matches_these_sources": same_src,
                      "matches_this_build": same_bin or same_src}
            qs = ("latent_rel_l2", "pixel_rel_l2_camera", "pixel_rel_l2_range")
            # a storage type MEETS the tolerance when every DDIM run measured for it does -- the DDIM-10 case and, where the suite
            # wrote it, the DDIM-50 case (every BASELINE configuration samples with DDIM); the shipped script's PLMS-50 at guidance
            # 5 is reported beside it, as are the runs of every case the file holds
            runs = {f"{case_key.split(' ')[0]}:{name}": {q: r[q] for q in qs}
                    for case_key, crec in pj.items() if isinstance(crec, dict)
                    for name, r in crec.items() if isinstance(r, dict) and all(q in r for q in qs)}
            ddim = lambda d: [v for k, v in runs.items() if k.split(":")[1] in (d, d + "_ddim50")]
            meets = {d: all(v[q] <= NORTH_STAR_REL_L2 for v in ddim(d) for q in qs) for d in ("bf16", "fp16") if ddim(d)}
            meets["over_tolerance"] = {k: {q: v[q] for q in qs if v[q] > NORTH_STAR_REL_L2} for k, v in runs.items()
                                       if any(v[q] > NORTH_STAR_REL_L2 for q in qs)}
            meets["runs"] = sorted(runs)
            meets["tolerance"] = NORTH_STAR_REL_L2
            meets["quantities"] = ("rel-L2 vs the CPU oracle of the final latent, the decoded camera picture and the decoded range view; "
                                   "bf16 / fp16: every DDIM run of the file (10 and 50 steps, both resolutions); *_plms50_cfg5: the shipped "
                                   "script's sampler, listed under over_tolerance when it is over")
            return {"parity": parity, "meets": meets}
        except (OSError, KeyError, StopIteration, ValueError, TypeError):
            continue
    return None


def build_model(workload, seed=0):
    """LatentDiffusion (UNet + camera VAE + lidar VAE) from configs/<workload>.yaml with random-init weights of
    the real architecture (also overwrites the zero-initialised layers).  No checkpoint exists offline: the lidar
    VAE's `ckpt_path` is cleared.  The conditioning producer (CLIP ViT-L/14 tower + mapper + box embedder) is built
    too: it is not part of a denoising step, but the end-to-end pass runs it as the harness does."""
    from mobi_amd.ldm.util import instantiate_from_config, load_config
    cfg = load_config(os.path.join(ROOT, "configs", f"{workload}.yaml"),
                      ["model.params.lidar_stage_config.params.ckpt_path=null"])
    assert cfg["latent_size"] == WORKLOADS[workload]["latent"]
    model = instantiate_from_config(cfg["model"])
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 1.0 / math.sqrt(p[0].numel()), generator=g)
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.normal_(0.0, 0.05, generator=g)
    return model.eval()


def spawn_ranks(n, argv):
    """`bench.py --gpus N` started as ONE process: launch N fresh ranks (one per GPU) as children of this
    process, which has not initialised the GPU (no exec of a GPU-holding process, ever), and pass their exit code on."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def stub_main(args, world, rank):
    """MOBI_BENCH_STUB=1: the launcher / timing / reduction / JSON contract with a CPU stand-in for the step
    (tests/test_bench_launcher_cpu.py drives `bench.py --gpus 2` through it on gloo, no GPU needed)."""
    import torch.distributed as dist
    torch.set_num_threads(1)                      # N ranks x the default intra-op pool oversubscribes a small host: the rank-dependent
    if world > 1:                                 # sleeps below, not thread contention, must decide the per-rank times
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("MOBI_BENCH_BACKEND", "gloo"))
    a = torch.randn(64, 64)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        a = torch.tanh(a @ a)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = torch.tanh(a @ a)
        time.sleep(0.002 * (1 + rank))           # rank-dependent: the MAX over ranks must win
    own = time.perf_counter() - t0               # this rank's own K steps (before it waits for the others)
    barrier()
    dt = time.perf_counter() - t0
    ranks = None
    if world > 1:
        tmax = torch.tensor([dt])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": "cpu (stub)",
                                       "ms_per_step": round(own / args.steps * 1e3, 3)})
    if rank == 0:
        elems = 16
        line = {"metric": "stub", "value": round(args.steps / dt * elems * world, 3), "unit": "stub element-forwards/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "stub (launcher test)", "parallelism": f"dp{world}"},
                "steps_per_s": round(args.steps / dt, 4)}
        if world > 1:
            line.update(ranks_seen=dist.get_world_size(), backend=dist.get_backend(), ranks=ranks,
                        ms_per_step_by_rank=[r["ms_per_step"] for r in ranks])
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mobi_nusc_512", choices=list(WORKLOADS))
    ap.add_argument("--objects", type=int, default=None, help="objects per GPU (default: the config's batch)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--cfg-scale", type=float, default=1.0, help="1.0 = harness default (no CFG)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write per-launch (kind, GFLOP, us) records of one step")
    ap.add_argument("--cpu-threads", type=int, default=32, help="threads for the CPU-oracle baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end objects/s pass")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from the host (no HIP graph)")
    ap.add_argument("--no-plms-line", action="store_true", help="skip the PLMS + CFG 5 (shipped invocation) leg")
    ap.add_argument("--no-fp16-line", action="store_true", help="skip the fp16-storage sub-record (the storage type that "
                    "meets the 1e-3 end-to-end tolerance; only added to a bf16 run)")
    ap.add_argument("--no-config-lines", action="store_true", help="skip the sub-records of BASELINE configs 2 and 5 "
                    "(`nusc256`: mobi_nusc_256, 4 objects, 32 x 32, DDIM-50, bf16; `ddim250_fp16`: the DDIM-250 / fp16 per-rank workload)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # nothing above has touched the GPU
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={os.environ['WORLD_SIZE']}", file=sys.stderr)
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("MOBI_BENCH_STUB") == "1":
        return stub_main(args, world, rank)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MOBI_BENCH_BACKEND=gloo + MOBI_BENCH_ONE_DEVICE=1 let the multi-rank logic be exercised on a 1-GPU box
    backend = os.environ.get("MOBI_BENCH_BACKEND", "nccl")
    if os.environ.get("MOBI_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        dist.init_process_group(backend=backend)         # "nccl" is RCCL on ROCm
    red_dev = device if backend == "nccl" else torch.device("cpu")

    import mobi_amd
    from mobi_amd import build, ops
    build.build(verbose=False)
    from mobi_amd.ldm.models.diffusion.ddim import DDIMSampler
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    mobi_amd.set_engine_dtype(dtype)

    wl = WORKLOADS[args.workload]
    B = args.objects or wl["objects"]
    side = wl["latent"]
    N = 2 * B                                           # camera/lidar interleaved
    cfg = args.cfg_scale != 1.0
    elems = N * (2 if cfg else 1)

    model = build_model(args.workload).to(device)
    net = model.model.diffusion_model
    sampler = DDIMSampler(model, graph=not args.no_graph)
    sampler.make_schedule(args.ddim_steps, ddim_eta=0.0, verbose=False)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    mk = lambda *s: torch.randn(*s, generator=g).to(device)
    img = mk(N, 4, side, side)
    inpaint = mk(N, 4, side, side)
    mask = torch.ones(N, 1, side, side)
    mask[:, :, side // 4: 3 * side // 4, side // 4: 3 * side // 4] = 0
    mask = mask.to(device)
    cond, uc = mk(N, 2, 768), mk(N, 2, 768)
    kw = {"test_model_kwargs": {"inpaint_image": inpaint, "inpaint_mask": mask}}
    total = sampler.ddim_timesteps.shape[0]
    steps_desc = list(reversed(sampler.ddim_timesteps.tolist()))

    def one_step(x, i):
        # what ddim_sampling does per iteration (ddim.py:141-161 of the reference): the int64 timestep vector, the
        # UNet evaluation(s), the fp32 update
        index = total - 1 - (i % total)
        step = int(steps_desc[i % total])
        ts = torch.full((N,), step, device=device, dtype=torch.long)
        x, _ = sampler.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=args.cfg_scale,
                                     unconditional_conditioning=uc if cfg else None, step_value=step, **kw)
        return x

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        x = img
        for i in range(args.warmup):
            x = one_step(x, i)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            x = one_step(x, args.warmup + i)
        ev1.record()
        t_issued = time.perf_counter() - t0            # host time to issue K steps (the GPU may still be running)
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0              # this rank's own K steps (before it waits for the others)
        barrier()
        dt = time.perf_counter() - t0
        gpu_ms_per_step = ev0.elapsed_time(ev1) / args.steps
        host_ms_per_step = t_issued / args.steps * 1e3
    ranks = None
    if world > 1:
        tmax = torch.tensor([dt], device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # evidence per rank for the scaling runs: which device every rank ran on, its own step time, and the world size
        # the collective backend itself reports
        prop = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "local_rank": local_rank, "device": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None),
                "uuid": str(getattr(prop, "uuid", "")), "ms_per_step": round(dt_own / args.steps * 1e3, 3),
                "gpu_ms_per_step": round(gpu_ms_per_step, 3)}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    finite = bool(torch.isfinite(x).all())

    # ---- the shipped invocation (scripts/realism_test_bench.sh:95-102): PLMS, classifier-free guidance 5 -------
    plms_line = None
    if not args.no_plms_line:
        from mobi_amd.ldm.models.diffusion.plms import PLMSSampler
        psampler = PLMSSampler(model, graph=not args.no_graph)
        psteps = 5 if args.steps >= 5 else 4             # S with 1000 % S == 0: exactly S sampler steps

        def plms_run(S):
            return psampler.sample(S=S, batch_size=N, shape=[4, side, side], conditioning=cond, verbose=False, x_T=img,
                                   unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
                                   inpaint_image=inpaint, inpaint_mask=mask)[0]

        with torch.no_grad():
            plms_run(2)                                    # 2 steps, 3 UNet evaluations: warm-up + graph capture
            plms_run(4)                                    # (capture leaves the GPU idle: 0.17 s of evaluations bring its clocks back)
            barrier()
            t0 = time.perf_counter()
            xs = plms_run(psteps)
            barrier()
            pdt = time.perf_counter() - t0
        nsteps = psampler.ddim_timesteps.shape[0]
        if world > 1:
            tmax = torch.tensor([pdt], device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            pdt = float(tmax.item())
        evals = nsteps + 1                                 # the first PLMS step evaluates the UNet twice (plms.py:226-232)
        plms_line = {"sampler": "plms", "cfg_scale": 5.0, "unet_batch": 2 * N, "sampler_steps": nsteps,
                     "unet_evaluations": evals, "ms_per_step": round(pdt / nsteps * 1e3, 3),
                     "value": round(evals * 2 * N * world / pdt, 3), "unit": "UNet element-forwards/s",
                     "finite": bool(torch.isfinite(xs).all())}

    # ---- end-to-end, the harness's own call sequence (scripts/inference_test_bench.py:416-464): get_input (four VAE
    # encodes, CLIP tower + mapper + box embedder for both modalities, the two reconstructions) -> DDIM -> decode_sample
    # -> log_data (post-processing + collages on the device) -> all-gather of the decoded samples -----------------
    objects_per_s = None
    if not args.no_e2e:
        from mobi_amd import dist as mdist
        R = side * 8
        hole = torch.ones(B, 1, R, R)
        hole[:, :, R // 4: 3 * R // 4, R // 4: 3 * R // 4] = 0
        x_T = mk(N, 4, side, side)

        def make_batch(seed):
            gi = torch.Generator(device="cpu").manual_seed(seed + rank)
            u = lambda *s_: torch.rand(*s_, generator=gi) * 2 - 1
            img_gt, rng_gt, ref = u(B, 3, R, R), u(B, 2, R, R), u(B, 3, 224, 224) * 1.5
            box = lambda: torch.rand(B, 8, 3, generator=gi)
            b = {"image": {"GT": img_gt, "inpaint_image": img_gt * hole, "inpaint_mask": hole.clone(),
                           "cond": {"ref_image": ref, "ref_bbox": box()}},
                 "lidar": {"range_data": rng_gt, "range_data_inpaint": rng_gt * hole, "range_mask": hole.clone(),
                           "range_instance_mask": (torch.rand(B, 1, R, R, generator=gi) > 0.8).float() * (1 - hole),
                           "min_depth_obj": torch.linspace(-0.8, -0.2, B), "max_depth_obj": torch.linspace(0.1, 0.7, B),
                           "width_crop": torch.full((B,), R // 2, dtype=torch.long),
                           "cond": {"ref_image": ref.clone(), "ref_bbox": box()}}}
            move = lambda d: {k: move(v) if isinstance(v, dict) else v.to(device) for k, v in d.items()}
            return move(b)

        gather_stat = {}

        phases = os.environ.get("MOBI_E2E_PHASES") == "1"         # development: device-synchronised phase times on stderr
        stamps = []

        def stamp(name):
            if phases:
                torch.cuda.synchronize()
                stamps.append((name, time.perf_counter()))

        def e2e(batch):
            stamp("start")
            data = model.get_input(batch, model.first_stage_key, force_c_encode=True, return_vae_rec=True)   # :416
            stamp("get_input")
            n = data["z"].shape[0]
            uc_ = torch.cat([model.learnable_vector.repeat(n, 1, 1), model.bbox_uncond_vector.repeat(n, 1, 1)], dim=1)
            smp, _ = sampler.sample(S=args.ddim_steps, batch_size=n, shape=[4, side, side], conditioning=data["cond"],
                                    verbose=False, eta=0.0, x_T=x_T, unconditional_guidance_scale=args.cfg_scale,
                                    unconditional_conditioning=uc_ if cfg else None,
                                    test_model_kwargs={"inpaint_image": data["z"][:, 4:8],
                                                       "inpaint_mask": data["z"][:, [8]]})                   # :447-461
            stamp("sample")
            h_cam, h_lid = model.decode_sample(smp, data.get("z_lidar"))                                      # :463
            stamp("decode_sample")
            log, _ = model.log_data(batch, data, h_cam, h_lid, log_metrics=False, return_sample=True, split="test")  # :464
            stamp("log_data")
            log = {k: log[k] for k in ("image_sample", "lidar_sample")}
            if backend != "nccl":
                log = {k: v.cpu() for k, v in log.items()}
            # the collective is timed with events on the stream it is issued on (no device sync inside the timed pass; the
            # gloo form moves host tensors and is timed on the host clock); only the TIMED pass's figure is reported
            if backend == "nccl":
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                res = mdist.gather_decoded(log, B * world)              # the one collective of the path
                ev[1].record()
                gather_stat["events"] = ev
            else:
                tg = time.perf_counter()
                res = mdist.gather_decoded(log, B * world)
                gather_stat["s"] = time.perf_counter() - tg
            gather_stat["bytes_per_rank"] = sum(v.numel() * v.element_size() for v in log.values())
            stamp("gather")
            if phases and rank == 0:
                print("e2e phases (ms): " + ", ".join(f"{b[0]} {1e3 * (b[1] - a_[1]):.1f}" for a_, b in zip(stamps, stamps[1:])),
                      file=sys.stderr)
            stamps.clear()
            return res

        with torch.no_grad():
            if args.warmup > 0:
                sampler_steps = args.ddim_steps
                args.ddim_steps = 2
                e2e(make_batch(99))                                     # short warm-up pass (weight packs, allocator)
                args.ddim_steps = sampler_steps
            # two timed passes over different images (no cached conditioning tokens): the FIRST still pays one-off costs the
            # short warm-up pass does not reach (allocator growth at DDIM-50's working set, first use of some code objects --
            # it read 5.97 against 6.6 objects/s on two boxes of the same build); the reported figure is the SECOND, the first
            # rides along as `objects_per_s_first_pass`
            e2e_first = None
            # Both batches exist BEFORE the first timed pass, as a harness with a prefetching DataLoader hands them over: building
            # a batch on the host between the passes leaves the GPU idle for ~0.1 s, its clocks drop, and whatever runs first
            # afterwards pays the ramp -- get_input 146 ... 206 ms instead of a steady 120 (`profiles/r05_e2e_idle_gap.txt`; neither
            # the cyclic collector nor the allocator: no device malloc happens in a steady pass).  MOBI_E2E_PREFETCH=0: the old order.
            seeds = (7, 8)
            pre = {sd_: make_batch(sd_) for sd_ in seeds} if os.environ.get("MOBI_E2E_PREFETCH", "1") == "1" else {}
            for seed in seeds:
                batch_t = pre[seed] if seed in pre else make_batch(seed)
                barrier()
                t0 = time.perf_counter()
                out = e2e(batch_t)
                barrier()
                e2e_dt = time.perf_counter() - t0
                if e2e_first is None:
                    e2e_first = e2e_dt
            if "events" in gather_stat:                                 # (read after the timed region's closing barrier)
                ev0, ev1 = gather_stat.pop("events")
                gather_stat["s"] = ev0.elapsed_time(ev1) * 1e-3
        if world > 1:
            tmax = torch.tensor([e2e_dt], device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            e2e_dt = float(tmax.item())
        assert out["image_sample"].shape == (B * world, 3, R, R) and out["lidar_sample"].shape == (B * world, 2, R, R)
        objects_per_s = B * world / e2e_dt
        if world > 1:
            tmax = torch.tensor([e2e_first], device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            e2e_first = float(tmax.item())
        objects_per_s_first = B * world / e2e_first
        sampler.make_schedule(args.ddim_steps, ddim_eta=0.0, verbose=False)

    roofline = None
    kinds = {}
    if not args.no_roofline and rank == 0:
        sink = []
        for rep in range(2):                     # first pass warms the event pool / allocator; the second is kept
            sink.clear()
            ops.set_profiler(sink)
            with torch.no_grad():
                one_step(x, args.warmup + args.steps + rep)
            torch.cuda.synchronize()
            ops.set_profiler(None)
        if args.dump_launches:
            with open(args.dump_launches, "w") as f:
                for kind, flops, e0, e1, nb, tag in sink:
                    f.write(f"{kind}\t{flops / 1e9:.4f}\t{e0.elapsed_time(e1) * 1e3:.2f}\t{nb / 1e6:.3f}\t{tag}\n")
        for kind, flops, e0, e1, nb, tag in sink:
            k = kinds.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0))
            k["launches"] += 1
            k["flops"] += flops
            k["ms"] += e0.elapsed_time(e1)
            k["bytes"] += nb
        ig = kinds["igemm"]
        # dominant kernel: the implicit-GEMM main loop that takes the most time of the step -- the library says which
        # variant every launch runs (mobi_igemm_kernel_variant): igemm_pp_kernel<...> (persistent direct-to-LDS,
        # ping-pong schedule), igemm_glds_kernel<...> (same geometry, lockstep) or igemm_kernel<...> (register-staged)
        # (a launch with a LayerNorm folded in -- tag suffix `_ln`, an instantiation of its own -- belongs to its tile geometry's
        #  main loop when the dominant kernel is chosen; `families` lists it apart)
        by_variant, by_variant_split = {}, {}
        for kind, flops, e0, e1, nb, tag in sink:
            if kind != "igemm":
                continue
            var = tag.split(" ")[0].replace("kern=", "") if tag.startswith("kern=") else "igemm"
            for table, key in ((by_variant, var[:-3] if var.endswith("_ln") else var), (by_variant_split, var)):
                d = table.setdefault(key, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0))
                d["launches"] += 1
                d["flops"] += flops
                d["ms"] += e0.elapsed_time(e1)
                d["bytes"] += nb
        dom = max(by_variant, key=lambda v: by_variant[v]["ms"]) if by_variant else "igemm"
        dk = by_variant.get(dom, ig)
        kname = {"pingpong": "igemm_pp_kernel", "direct_lds": "igemm_glds_kernel", "staged128": "igemm_kernel",
                 "staged256": "igemm_kernel", "ring128": "igemm_ring_kernel", "ring256": "igemm_ring_kernel", "ring128w": "igemm_ring_kernel",
                 "ring256_ln": "igemm_ring_kernel", "ring128_ln": "igemm_ring_kernel",
                 "small": "small_gemm_kernel"}.get(dom, "igemm_kernel")
        ach = dk["flops"] / (dk["ms"] * 1e-3) / 1e12
        # which roof bounds the kernel's launches of this step taken together: their matrix time at the dense peak against
        # their algorithmic bytes at the HBM peak (both ideal); MFMA for every variant on these workloads
        t_mfma, t_hbm = dk["flops"] / (PEAK_TFLOPS * 1e12), dk["bytes"] / 8.0e12
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        peak, unit = (PEAK_TFLOPS, "TFLOP/s") if bound == "mfma" else (8000.0, "GB/s")
        if bound == "hbm":
            ach = dk["bytes"] / (dk["ms"] * 1e-3) / 1e9
        roofline = {"bound": bound, "kernel": kname, "achieved": round(ach, 2), "peak": peak,
                    "unit": unit, "frac": round(ach / peak, 4), "traffic": None,
                    "launches_per_step": dk["launches"],
                    "avg_launch_us": round(dk["ms"] * 1e3 / dk["launches"], 2),
                    "gflop_per_launch": round(dk["flops"] / dk["launches"] / 1e9, 3),
                    "ms_per_step": round(dk["ms"], 3),
                    "note": "HIP-event-bracketed launches (on the launch stream) of one extra step after the timed "
                            "region; algorithmic 2*M*N*K of every conv / linear the kernel ran"}
        roofline["algorithmic_mb_per_launch"] = round(dk["bytes"] / dk["launches"] / 1e6, 2)
        roofline["ideal_ms_mfma_vs_hbm"] = [round(t_mfma * 1e3, 3), round(t_hbm * 1e3, 3)]
        roofline["igemm_ms_by_variant"] = {v: round(d["ms"], 3) for v, d in by_variant.items()}
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of THIS command (tools/profile_round.sh);
        # the committed figure is attached only when it was collected on the same workload / batch / dtype / guidance
        pmc = os.path.join(ROOT, "profiles", PMC_TRAFFIC)
        run_cfg = {"workload": args.workload, "objects": B, "dtype": args.dtype, "cfg_scale": args.cfg_scale}
        if os.path.exists(pmc):
            with open(pmc) as f:
                doc = json.load(f)
            rec = doc.get(kname)
            if rec and doc.get("config") == run_cfg:
                roofline["traffic"] = round(rec["hbm_bytes_per_launch_corrected"])
                roofline["traffic_source"] = f"profiles/{PMC_TRAFFIC} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
            else:
                roofline["traffic_source"] = f"none: profiles/{PMC_TRAFFIC} was collected on {doc.get('config')}"
        roofline["igemm_all_tflops"] = round(ig["flops"] / (ig["ms"] * 1e-3) / 1e12, 2)
        roofline["igemm_all_launches"] = ig["launches"]
        if "attention" in kinds:
            at = kinds["attention"]
            roofline["attention_tflops"] = round(at["flops"] / (at["ms"] * 1e-3) / 1e12, 2)
            roofline["attention_ms_per_step"] = round(at["ms"], 3)
        roofline["igemm_ms_per_step"] = round(ig["ms"], 3)
        roofline["norm_ms_per_step"] = round(sum(kinds.get(k, {"ms": 0})["ms"] for k in ("groupnorm", "layernorm")), 3)
        # every family of the step against ITS roof (matrix families: algorithmic FLOP/s over the dense peak; bandwidth
        # families: algorithmic bytes/s over 8 TB/s), and the step if every launch ran at the better of its two roofs
        fam = {}
        for v, d in by_variant_split.items():
            fam[f"igemm_{v}"] = {"ms": round(d["ms"], 3), "launches": d["launches"],
                                 "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 1),
                                 "frac": round(d["flops"] / (d["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS, 4)}
        for kname_, d in kinds.items():
            if kname_ == "igemm" or d["ms"] <= 0:
                continue
            if d["flops"] > 0:
                fam[kname_] = {"ms": round(d["ms"], 3), "launches": d["launches"],
                               "tflops": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 1),
                               "frac": round(d["flops"] / (d["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS, 4)}
            else:
                fam[kname_] = {"ms": round(d["ms"], 3), "launches": d["launches"],
                               "gb_s": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                               "frac": round(d["bytes"] / (d["ms"] * 1e-3) / 8.0e12, 4)}
        roofline["families"] = fam
        ideal = sum(max(flops / (PEAK_TFLOPS * 1e12), nb / 8.0e12) for _, flops, _, _, nb, _ in sink)
        roofline["ms_per_step_every_launch_at_its_better_roof"] = round(ideal * 1e3, 3)
        roofline["ms_per_step_sum_of_launches"] = round(sum(e0.elapsed_time(e1) for _, _, e0, e1, _, _ in sink), 3)
        roofline["launches_per_step_all"] = len(sink)
        # split-K launches of the step: how many leave their slabs to the GroupNorm that reads the result (ops.Deferred) and how
        # many are followed by a reduce launch of their own (inside their igemm bracket, or a later mobi_igemm_finish)
        n_split = sum(1 for kind, *_, tag in sink if kind == "igemm" and " split=" in tag and " split=1 " not in tag)
        n_gn = sum(1 for kind, *_, tag in sink if kind == "groupnorm" and " slabs=" in tag)
        roofline["split_k_launches"] = {"all": n_split, "summed_by_the_consuming_groupnorm": n_gn, "reduce_launches": n_split - n_gn}
        if os.path.exists(pmc):
            traffic = {}
            for fam_k in ("igemm_ring_kernel", "igemm_pp_kernel", "attention_rows_kernel", "attention_kernel", "ff_geglu_kernel",
                          "gn_regs_kernel", "gn_stats_kernel", "gn_apply_kernel", "layernorm_kernel", "row_chain_kernel"):
                rec = doc.get(fam_k)
                if rec and doc.get("config") == run_cfg:
                    traffic[fam_k] = round(rec["hbm_bytes_per_launch_corrected"])
            if traffic:
                roofline["traffic_by_kernel"] = traffic

    cpu_baseline = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import unet as ounet
        ncpu = max(1, min(os.cpu_count() or 1, args.cpu_threads))   # more threads than this slows torch's CPU convs down
        torch.set_num_threads(ncpu)
        sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
        xc = torch.cat([img[:2], inpaint[:2], mask[:2]], 1).float().cpu()
        tc = torch.full((2,), 500, dtype=torch.long)
        cc = cond[:2].float().cpu()
        with torch.no_grad():
            ounet.unet_forward(sd, ounet.UNetConfig(), xc, tc, cc)      # untimed: thread pool, allocator, page-in
            t0 = time.perf_counter()
            ounet.unet_forward(sd, ounet.UNetConfig(), xc, tc, cc)
            cdt = time.perf_counter() - t0
        cpu_baseline = {"value": round(2 / cdt, 4), "unit": "UNet element-forwards/s", "cores": ncpu, "kind": "port",
                        "sample": f"1 denoising step of 1 object (UNet batch 2) at latent {side}x{side}, fp32, "
                                  f"PyTorch CPU oracle of the reference graph, {cdt:.1f} s (second of two evaluations)"}
        del sd

    # ---- the same step with fp16 storage: the storage type that meets the 1e-3 end-to-end tolerance ----------
    fp16_line = None
    if args.dtype == "bf16" and not args.no_fp16_line:
        mobi_amd.set_engine_dtype(torch.float16)
        s16 = DDIMSampler(model, graph=not args.no_graph)
        s16.make_schedule(args.ddim_steps, ddim_eta=0.0, verbose=False)

        def step16(x_, i):
            index = total - 1 - (i % total)
            step = int(steps_desc[i % total])
            ts = torch.full((N,), step, device=device, dtype=torch.long)
            return s16.p_sample_ddim(x_, cond, ts, index=index, unconditional_guidance_scale=args.cfg_scale,
                                     unconditional_conditioning=uc if cfg else None, step_value=step, **kw)[0]

        with torch.no_grad():
            x16 = img
            for i in range(max(args.warmup, SUB_WARMUP_STEPS)):
                x16 = step16(x16, i)
            barrier()
            t0 = time.perf_counter()
            for i in range(args.steps):
                x16 = step16(x16, args.warmup + i)
            barrier()
            dt16 = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt16], device=red_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt16 = float(tmax.item())
        fp16_line = {"dtype": "fp16", "steps": args.steps, "ms_per_step": round(dt16 / args.steps * 1e3, 3),
                     "value": round(args.steps / dt16 * elems * world, 3), "unit": "UNet element-forwards/s",
                     "finite": bool(torch.isfinite(x16).all())}
        del s16
        mobi_amd.set_engine_dtype(dtype)

    # ---- BASELINE configs 2 and 5 on the same line (the driver times only this invocation) ---------------------------------
    # config 2: configs/mobi_nusc_256.yaml, batch 4, DDIM-50, bf16 -- the UNet of that config IS this one (the two YAML files
    # differ in image_size / latent_size only: checked), so the 32 x 32 step runs on the same weights;
    # config 5: mobi_nusc_all-classes_512 + range_autoencoder, DDIM-250, fp16 -- same model as config 3 (the configs differ in the
    # dataset's classes), 8 objects per rank, the [1, 5, ..., 997] table; the step time does not depend on the timestep, so a
    # handful of steps of the 250 are timed.
    config_lines = {}
    if args.workload == "mobi_nusc_512" and not args.no_config_lines and not cfg:
        from mobi_amd.ldm.util import load_config

        def unet_params(name):
            c = load_config(os.path.join(ROOT, "configs", name), ["model.params.lidar_stage_config.params.ckpt_path=null"])
            u = dict(c["model"]["params"]["unet_config"]["params"])
            u.pop("image_size", None)
            return u

        def timed_steps(smp, n_el, sd, S, label):
            gg = torch.Generator(device="cpu").manual_seed(4321 + rank)
            mk2 = lambda *s_: torch.randn(*s_, generator=gg).to(device)
            x_, inp_ = mk2(n_el, 4, sd, sd), mk2(n_el, 4, sd, sd)
            m_ = torch.ones(n_el, 1, sd, sd)
            m_[:, :, sd // 4: 3 * sd // 4, sd // 4: 3 * sd // 4] = 0
            m_ = m_.to(device)
            c_ = mk2(n_el, 2, 768)
            kw_ = {"test_model_kwargs": {"inpaint_image": inp_, "inpaint_mask": m_}}
            tot = smp.ddim_timesteps.shape[0]
            desc = list(reversed(smp.ddim_timesteps.tolist()))

            def stp(x__, i):
                index = tot - 1 - (i % tot)
                step = int(desc[i % tot])
                ts = torch.full((n_el,), step, device=device, dtype=torch.long)
                return smp.p_sample_ddim(x__, c_, ts, index=index, step_value=step, **kw_)[0]

            with torch.no_grad():
                for i in range(max(args.warmup, SUB_WARMUP_STEPS * (4 if sd < 64 else 1))):
                    x_ = stp(x_, i)
                barrier()
                t0_ = time.perf_counter()
                for i in range(args.steps):
                    x_ = stp(x_, args.warmup + i)
                barrier()
                d_ = time.perf_counter() - t0_
            if world > 1:
                tm = torch.tensor([d_], device=red_dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d_ = float(tm.item())
            return d_, bool(torch.isfinite(x_).all()), [int(v) for v in smp.ddim_timesteps[:3]] + [int(smp.ddim_timesteps[-1])]

        def timed_in_flight(smps, n_el, sd):
            """Every sampler of `smps` owns a request (its own latents, conditioning and step graph; the weights are shared) and
            a stream; a step = one denoising step of EVERY request, issued round-robin.  What a server with several requests
            queued gets out of the card: a second stream's launches fill the ramp / drain of the first's."""
            gg = torch.Generator(device="cpu").manual_seed(8765 + rank)
            mk2 = lambda *s_: torch.randn(*s_, generator=gg).to(device)
            reqs = []
            for smp in smps:
                m_ = torch.ones(n_el, 1, sd, sd)
                m_[:, :, sd // 4: 3 * sd // 4, sd // 4: 3 * sd // 4] = 0
                reqs.append({"smp": smp, "x": mk2(n_el, 4, sd, sd), "c": mk2(n_el, 2, 768), "st": torch.cuda.Stream(device=device),
                             "kw": {"test_model_kwargs": {"inpaint_image": mk2(n_el, 4, sd, sd), "inpaint_mask": m_.to(device)}}})
            tot = smps[0].ddim_timesteps.shape[0]
            desc = list(reversed(smps[0].ddim_timesteps.tolist()))

            def stp_all(i):
                index = tot - 1 - (i % tot)
                step = int(desc[i % tot])
                for r in reqs:
                    with torch.cuda.stream(r["st"]):
                        ts = torch.full((n_el,), step, device=device, dtype=torch.long)
                        r["x"] = r["smp"].p_sample_ddim(r["x"], r["c"], ts, index=index, step_value=step, **r["kw"])[0]

            with torch.no_grad():
                for r in reqs:
                    r["st"].wait_stream(torch.cuda.current_stream())
                for i in range(max(args.warmup, SUB_WARMUP_STEPS * (4 if sd < 64 else 1))):
                    stp_all(i)
                barrier()
                t0_ = time.perf_counter()
                for i in range(args.steps):
                    stp_all(args.warmup + i)
                barrier()
                d_ = time.perf_counter() - t0_
            if world > 1:
                tm = torch.tensor([d_], device=red_dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d_ = float(tm.item())
            return d_, all(bool(torch.isfinite(r["x"]).all()) for r in reqs)

        def in_flight_record(n_el, sd, gf, what):
            smps = []
            for _ in range(2):
                s_ = DDIMSampler(model, graph=not args.no_graph)
                s_.make_schedule(50, ddim_eta=0.0, verbose=False)
                smps.append(s_)
            d_, fin_ = timed_in_flight(smps, n_el, sd)
            v_ = args.steps / d_ * 2 * n_el * world
            return {"config": f"NOT a BASELINE configuration -- TWO independent requests of {what} in flight on two streams (own latents, "
                              f"conditioning and step graphs, shared weights): the card's throughput when requests queue up",
                    "dtype": args.dtype, "steps": args.steps, "requests_in_flight": 2,
                    "ms_per_step_of_both": round(d_ / args.steps * 1e3, 3), "ms_per_step_per_request": round(d_ / args.steps * 1e3 / 2, 3),
                    "value": round(v_, 3), "unit": "UNet element-forwards/s", "model_tflops": round(v_ * gf / 1e3 / world, 2),
                    "model_frac_of_peak": round(v_ * gf / 1e3 / world / PEAK_TFLOPS, 4), "finite": fin_}

        if not args.no_graph:
            config_lines["in_flight2"] = in_flight_record(N, side, wl["gflop_per_element"] - wl["gflop_skippable"],
                                                          f"this line's batch ({B} objects, UNet batch {N})")
        if unet_params("mobi_nusc_256.yaml") == unet_params("mobi_nusc_512.yaml"):
            w2 = WORKLOADS["mobi_nusc_256"]
            s256 = DDIMSampler(model, graph=not args.no_graph)
            s256.make_schedule(50, ddim_eta=0.0, verbose=False)
            n2 = 2 * w2["objects"]
            d2, fin2, _ = timed_steps(s256, n2, w2["latent"], 50, "nusc256")
            v2 = args.steps / d2 * n2 * world
            gf2 = w2["gflop_per_element"] - w2["gflop_skippable"]
            config_lines["nusc256"] = {
                "config": "BASELINE configs[1]: mobi_nusc_256, 4 objects/GPU (UNet batch 8), latent 32x32, DDIM-50 schedule, cfg_scale 1.0",
                "dtype": args.dtype, "steps": args.steps, "ms_per_step": round(d2 / args.steps * 1e3, 3), "value": round(v2, 3),
                "unit": "UNet element-forwards/s", "model_tflops": round(v2 * gf2 / 1e3 / world, 2),
                "model_frac_of_peak": round(v2 * gf2 / 1e3 / world / PEAK_TFLOPS, 4), "finite": fin2}
            del s256
            if not args.no_graph:
                config_lines["nusc256_in_flight2"] = in_flight_record(n2, w2["latent"], gf2, "configs[1]'s batch (4 objects, UNet batch 8)")
        mobi_amd.set_engine_dtype(torch.float16)
        s250 = DDIMSampler(model, graph=not args.no_graph)
        s250.make_schedule(250, ddim_eta=0.0, verbose=False)
        d5, fin5, tab5 = timed_steps(s250, N, side, 250, "ddim250_fp16")
        v5 = args.steps / d5 * N * world
        config_lines["ddim250_fp16"] = {
            "config": f"BASELINE configs[4], per rank: mobi_nusc_all-classes_512 (= the mobi_nusc_512 model), {B} objects/GPU (UNet batch "
                      f"{N}), latent {side}x{side}, DDIM-250 schedule, fp16 storage, cfg_scale 1.0",
            "dtype": "fp16", "steps": args.steps, "ddim_timesteps_first3_last": tab5,
            "ms_per_step": round(d5 / args.steps * 1e3, 3), "value": round(v5, 3), "unit": "UNet element-forwards/s",
            "sampling_s_per_batch_ddim250": round(d5 / args.steps * 250, 3),
            "model_frac_of_peak": round(v5 * (wl["gflop_per_element"] - wl["gflop_skippable"]) / 1e3 / world / PEAK_TFLOPS, 4),
            "finite": fin5}
        del s250
        mobi_amd.set_engine_dtype(dtype)

    if rank == 0:
        steps_per_s = args.steps / dt
        value = steps_per_s * elems * world
        useful_gf = wl["gflop_per_element"] - wl["gflop_skippable"]
        out = {
            "metric": "denoising throughput (UNet element-forwards/s = steps/s x UNet batch)",
            "value": round(value, 3), "unit": "UNet element-forwards/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} UNet denoising step, {B} objects/GPU (UNet batch {elems}), "
                                   f"latent {side}x{side}, DDIM-{args.ddim_steps} schedule, "
                                   f"cfg_scale {args.cfg_scale}",
                       "objects_per_gpu": B, "unet_batch": elems, "latent": side, "sampler": "ddim",
                       "parallelism": f"dp{world} (objects sharded, no collective in the loop)"},
            "steps_per_s": round(steps_per_s, 4),          # denoising steps per second of every rank's own batch
            "host_ms_per_step": round(host_ms_per_step, 3), "gpu_ms_per_step": round(gpu_ms_per_step, 3),
            "step_graph": bool(not args.no_graph),
            "model_tflops": round(value * useful_gf / 1e3 / world, 2),
            "model_frac_of_peak": round(value * useful_gf / 1e3 / world / PEAK_TFLOPS, 4),
            "finite": finite,
        }
        if objects_per_s is not None:
            out["objects_per_s"] = round(objects_per_s, 4)
            out["objects_per_s_first_pass"] = round(objects_per_s_first, 4)
            out["e2e"] = (f"{B} objects/GPU, the harness's calls (inference_test_bench.py:416-464): get_input (4 VAE encodes, CLIP "
                          f"ViT-L/14 tower + mapper + box embedder, 2 reconstruction decodes) + DDIM-{args.ddim_steps} + decode_sample + "
                          f"log_data (2 VAE decodes, range de-normalisation, uint8 collages on the device) + all-gather of the decoded samples")
        if plms_line:
            out["plms_cfg5"] = plms_line
        if fp16_line:
            out["fp16"] = fp16_line
        out.update(config_lines)
        # end-to-end parity at production width, from the committed measurement of the GPU suite (never re-measured here: the
        # oracle run behind it takes minutes of CPU): both storage types, the benched one first
        par = parity_record(side, args.dtype)
        if par:
            out["parity"] = par["parity"]
            out["meets_north_star_tolerance"] = par["meets"]
        out["headline_frac_of_peak"] = out["model_frac_of_peak"]     # the model-level fraction; roofline.frac is one kernel's
        if world > 1:
            out["ranks_seen"] = dist.get_world_size()
            out["backend"] = dist.get_backend()
            out["ranks"] = ranks
            out["ms_per_step_by_rank"] = [r["ms_per_step"] for r in ranks]
        if objects_per_s is not None and gather_stat:
            out["all_gather"] = {"bytes_per_rank": int(gather_stat["bytes_per_rank"]),
                                 "bytes_total": int(gather_stat["bytes_per_rank"]) * world,
                                 "ms": round(gather_stat["s"] * 1e3, 3)}
        if roofline:
            out["roofline"] = roofline
        if cpu_baseline:
            out["cpu_baseline"] = cpu_baseline
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
