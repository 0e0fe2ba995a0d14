#!/usr/bin/env python3
"""Copy the summaries of one `tools/profile_round.sh <tag>` call from gpurun_out/<tag>/ into profiles/ and regenerate
profiles/<tag>_README.md from them (kernel table from the rocprofv3 stats, dominant-kernel cross-check against the
bench line's own HIP-event brackets, PMC traffic, the other bench lines of the same build).

    python tools/collect_profiles.py r02
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line_of(path):
    with open(path) as f:
        for l in f:
            if l.startswith("{"):
                return json.loads(l)
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")

    def cp(a, b):
        if os.path.exists(os.path.join(src, a)):
            shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))

    cp("kernel_stats.csv", "unet512_b16_bf16_kernel_stats.csv")
    cp("bench.json", "bench_unet512_b16_bf16.json")
    cp("launches.tsv", "launches_one_step.tsv")
    cp("launches256.tsv", "launches_one_step_nusc256.tsv")
    cp("pmc_traffic.json", "pmc_traffic.json")
    cp("vae_prof.txt", "vae_encode_decode.txt")
    cp("chain_lab.txt", "chain_lab.txt")
    cp("chain_stamps.txt", "chain_stamps.txt")
    for name in ("graph_breakdown.txt", "mall_share.txt", "vae_decode_fp16.txt", "nusc256_kernel_stats.csv", "train_kernel_stats.csv",
                 "nusc256_pmc_traffic.json", "train_step.txt"):        # tools/profile_more.sh
        cp(name, name)
    # end-to-end parity as the GPU suite of the same build last measured it (tests/test_gpu_production.py writes
    # gpurun_out/parity_last.json, tied to the library's hash): kept with the commit it was measured at
    plast = os.path.join(ROOT, "gpurun_out", "parity_last.json")
    if os.path.exists(plast):
        import subprocess as sp
        doc = json.load(open(plast))
        doc["commit"] = sp.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        if "src_sha16" not in doc:
            # a record written before the suite stamped the sources: the local library IS the measured one (same binary hash) and is
            # current for the local sources, so their hash is the record's
            import hashlib
            sys.path.insert(0, ROOT)
            from mobi_amd import build
            with open(build.LIB, "rb") as f:
                local = hashlib.sha256(f.read()).hexdigest()[:16]
            if local == doc.get("lib_sha16") and build.up_to_date():
                doc["src_sha16"] = build.sources_sha16()
        with open(os.path.join(dst, f"{tag}_parity.json"), "w") as f:
            json.dump(doc, f, indent=1)
    bench = line_of(os.path.join(src, "bench.json"))
    pmc = json.load(open(os.path.join(src, "pmc_traffic.json")))
    rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
    # fully demangled kernel names (rocprofv3 leaves some instances mangled, some half-demangled)
    import subprocess
    filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if os.path.exists(filt):
        names = subprocess.run([filt], input="\n".join(r["Name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
        for r, nm in zip(rows, names):
            r["Name"] = nm.replace("__bf16", "bf16").replace("_Float16", "f16")
        with open(os.path.join(dst, f"{tag}_unet512_b16_bf16_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    roof = bench["roofline"]
    kname = roof["kernel"]
    dom = [r for r in rows if kname in r["Name"]]
    # igemm_ring_kernel has two geometries in one name: the dominant variant is the eight-wave 256 x 320 (256) tile (<..., 8, 8>),
    # the four-wave 128 x 160 tile (<..., 4, 4>, the small-m launches) is another family of the bench line
    var = max(roof.get("igemm_ms_by_variant", {"": 0}), key=lambda v: roof.get("igemm_ms_by_variant", {"": 0})[v])
    if kname == "igemm_ring_kernel" and var in ("ring256", "ring128", "ring256_ln", "ring128_ln"):
        # (template arguments: ..., NW, MT, LNF -- the LayerNorm-folded instantiations are kernels of their own)
        want = (", 8, 8, " if var.startswith("ring256") else ", 4, 4, ") + ("true>" if var.endswith("_ln") else "false>")
        dom = [r for r in dom if want in r["Name"]] or dom
    calls = sum(int(r["Calls"]) for r in dom)
    tot_ns = sum(float(r["TotalDurationNs"]) for r in dom)
    lines = [f"# Profiles of round {tag[1:].lstrip('0')} (MI355X, 1 GPU) -- tag {tag}", "",
             f"All files come from ONE GPU-box call of `bash tools/profile_round.sh {tag}` (bench line, launch dump, rocprofv3",
             "`--kernel-trace --stats`, one `--pmc` pass per counter, and the other bench lines of the same build), copied here by",
             "`tools/collect_profiles.py`.  The traced runs issue every launch from the host (`--no-graph`): the same kernels, one",
             "dispatch record each; the bench lines replay the captured step graph.", "",
             f"Workload: {bench['config']['workload']}, {bench['dtype']}; 7 steps in the trace.", "",
             f"Bench line of the same box: {bench['value']:.1f} {bench['unit']}, {bench['ms_per_step']:.2f} ms per step "
             f"(host {bench['host_ms_per_step']:.2f} ms to issue a step, GPU {bench['gpu_ms_per_step']:.2f} ms), model rate "
             f"{bench['model_tflops']:.0f} TFLOP/s ({100 * bench['model_frac_of_peak']:.1f} % of the bf16 MFMA peak), "
             f"{bench.get('objects_per_s', 0):.2f} inpainted objects/s end to end"
             + (f", CPU baseline {bench['cpu_baseline']['value']:.3f} {bench['cpu_baseline']['unit']} on "
                f"{bench['cpu_baseline']['cores']} threads" if bench.get("cpu_baseline") else "") + ".", ""]
    others = []
    for fn, what in (("bench_fp16.json", "fp16 storage"), ("bench_cfg5.json", "DDIM, classifier-free guidance 5 (UNet batch 32)"),
                     ("bench_256.json", "BASELINE config 2: mobi_nusc_256, 4 objects (UNet batch 8), step graph"),
                     ("bench_256_nograph.json", "the same, every launch issued from the host")):
        d = line_of(os.path.join(src, fn)) if os.path.exists(os.path.join(src, fn)) else None
        if d:
            others.append(f"| {what} | {d['ms_per_step']:.2f} | {d['value']:.1f} | {d.get('host_ms_per_step', 0):.2f} | "
                          f"{d.get('plms_cfg5', {}).get('value', '')} |")
    if bench.get("plms_cfg5"):
        p = bench["plms_cfg5"]
        others.insert(0, f"| shipped invocation: PLMS, guidance 5, {p['unet_evaluations']} UNet evaluations of batch {p['unet_batch']} "
                         f"| {p['ms_per_step']:.2f} | {p['value']:.1f} | | |")
    if others:
        lines += ["| other lines of the same build / box | ms per step | UNet element-forwards/s | host ms per step | PLMS+CFG5 el-fwd/s |",
                  "|---|---|---|---|---|"] + others + [""]
    lines += ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:18]:
        lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                     f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines += ["",
              f"Dominant kernel = `{kname}<...>` ({len(dom)} template instances in the table): {calls} calls, {tot_ns / 1e6:.1f} ms, average "
              f"{tot_ns / max(calls, 1) / 1e3:.1f} us under rocprofv3 vs `roofline.avg_launch_us` = {roof['avg_launch_us']} us from bench.py's own "
              f"HIP-event brackets on the same box.  `roofline.achieved` = {roof['achieved']} TFLOP/s = {roof['gflop_per_launch']} GFLOP "
              f"(algorithmic 2*M*N*K) / that launch time; frac {roof['frac']}.  igemm time per step by main loop: "
              f"{roof['igemm_ms_by_variant']}.", "`__amd_rocclr_copyBuffer` rows are the one-off weight upload / packing, not per-step work.", ""]
    for fam in ("igemm_pp_kernel", "igemm_ring_kernel", "attention_rows_kernel", "attention_kernel", "gn_regs_kernel", "row_chain_kernel",
                "ff_geglu_kernel"):
        rec = pmc.get(fam)
        if rec:
            lines.append(f"HBM traffic of `{fam}` (`{tag}_pmc_traffic.json`, FETCH_SIZE x 2 + WRITE_SIZE, separate `--pmc` passes): "
                         f"{rec['hbm_bytes_per_launch_corrected'] / 1e6:.1f} MB per launch over {rec['launches']} launches.")
    ms = bench["ms_per_step"] if bench else None
    group = "" if ms is None else (" (this set: %.2f ms per step -- the %s group)" % (ms, "fast" if ms < 19.8 else "slow"))
    lines += ["", "Box spread: the pool's boxes fall into two groups about 6 % apart (clocks under load): 19.2-19.5 ms per step on the fast "
              f"ones, 20.1-20.9 on the slow ones for the same build{group}; every A/B in this directory is interleaved on one box.",
              "", "Files of this tag:"]
    known = [("unet512_b16_bf16_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of the traced bench run (bf16, every launch host-issued)"),
             ("bench_unet512_b16_bf16.json", "bench line of the same box"),
             ("launches_one_step.tsv", "kind, GFLOP, us, algorithmic MB, kernel variant and shape of every launch of one mobi_nusc_512 step"),
             ("launches_one_step_nusc256.tsv", "the same of one mobi_nusc_256 step"),
             ("pmc_traffic.json", "per-kernel-family FETCH_SIZE / WRITE_SIZE, separate --pmc passes (tools/pmc_summary.py)"),
             ("vae_encode_decode.txt", "per-launch times of the VAEs (8 images, 512 x 512, both autoencoders; tools/vae_prof.py)"),
             ("vae_trunk.txt", "the VAE decoders with the fp32 residual trunk"),
             ("chain_lab.txt", "the row-chain kernel against the launches it replaces"),
             ("chain_stamps.txt", "its phase stamps and ablations"),
             ("error_table.txt", "measured rel-L2 of every parity assertion of `pytest -m gpu` (MOBI_RECORD_ERRORS, tools/error_table.py)"),
             ("parity.json", "end-to-end parity at production width (latent / pixel space, both storage types)"),
             ("graph_breakdown.txt", "per-kernel time INSIDE the replayed step graph (tools/graph_gaps.py)"),
             ("nusc256_kernel_stats.csv", "rocprofv3 stats of the mobi_nusc_256 step, every launch host-issued (tools/profile_extra.sh)"),
             ("nusc256_pmc_traffic.json", "FETCH_SIZE / WRITE_SIZE per kernel family of the mobi_nusc_256 step (small_gemm_kernel: 11.4 MB per launch)"),
             ("train_kernel_stats.csv", "rocprofv3 stats of full-width training steps (tools/train_bench.py via tools/profile_extra.sh)"),
             ("train_step.txt", "training-step timings, change by change"),
             ("small_lab.txt", "the small-problem igemm kernel against the LDS-ring kernels per launch (tools/small_lab.py)"),
             ("small_lab_conv.txt", "its 3 x 3 form (not routed: slower)"),
             ("small_lab_v1_fragment_loads.txt", "its first form (fragment-shaped global loads), for the record"),
             ("ab_small.txt", "whole-step A/B of the small-problem kernel (tools/ab_small.sh)"),
             ("ab_split_longk.txt", "whole-step A/B of the long-k split-K rule"), ("ab_attn_nw8.txt", "whole-step A/B of the 8-wave attention block rule"),
             ("splitk_sweep_512.txt", "split-K sweep of the mobi_nusc_512 step's small-m shapes on the current build (tools/sweep_split.py)"),
             ("splitk_sweep_256.txt", "the same for mobi_nusc_256"),
             ("ab_attn_h16.txt", "A/B of the 16x16x32 P.V form of dh = 40 attention (tools/ab_attn_h16.sh: slower, off by default)"),
             ("ab_row_chain.txt", "whole-step A/B of the row chains"), ("ab_prechain.txt", "the pre-attention chain (slower, off by default)"),
             ("fused_split_lab.txt", "split-K finished inside the launch against slabs + reduce launch (tools/fsplit_lab.py: slower, off by default)"),
             ("ab_grouped_q_ln_fold.txt", "whole-step A/B of the grouped cross-modal to_q launch and of the LayerNorm fold (tools/ab_cfg.sh)"),
             ("ln_fold_launches.txt", "per-launch times with the LayerNorm fold on / off"),
             ("gn_lab.txt", "GroupNorm forms per shape incl. the chunked one-launch kernel (tools/gn_lab.py)"),
             ("decoder_err.txt", "the decoders' own error on the oracle's latent, by precision option (tests/decoder_err.py)"),
             ("conc_lab.txt", "two independent half batches on two streams against one batch (tools/conc_lab.py)"),
             ("mall_share.txt", "Infinity-Cache share of the ring kernel's counter traffic: the step's launch against one scaled past 256 MiB (tools/mall_share.sh)"),
             ("vae_decode_fp16.txt", "fp16 VAE decode / encode of 8 images at 512 x 512 by precision option (fp32 trunk / streams / hi | lo tail)"),
             ("ab_ln_fold.txt", "whole-step A/B of the shipped LayerNorm fold, both workloads (slow-group box)"),
             ("ab_chain_ff.txt", "row chains / one-launch feed-forward re-measured against the one-by-one sequences as they are now"),
             ("tile_geometry_lab.txt", "the small-m 1 x 1 shapes on every tile geometry: ~10 us of fixed cost per launch (tools/wide_lab.py)"),
             ("ab_defer_split.txt", "split-K slabs summed by the consuming GroupNorm against the reduce launches: parity tests, then the whole-step A/B (tools/ab_defer.sh)"),
             ("ab_runtime_env.txt", "HIP runtime launch switches on the replayed mobi_nusc_256 graph (nothing to gain)"),
             ("ab_fp32_outer_stream.txt", "the UNet's outer residual stream carried in fp32 beside its 16-bit copy: not where the fp16 error comes from"),
             ("ab_graph_branches.txt", "the ResBlocks' 1 x 1 skip convolution on a forked branch of the step graph (slower: fork / join edges)"),
             ("ab_ln_fold_rows.txt", "the LayerNorm fold's row threshold (LN_FOLD_MIN_ROWS), whole-step A/B"),
             ("ab_split_round4.txt", "round 5's split-K plan against round 4's, whole-step A/B"),
             ("split_sweep.txt", "split-K sweeps of both workloads with the reduce launch timed in (tools/sweep_split.py)"),
             ("attn_bwd_err.txt", "the softmax backward's row term: D from the stored output against D = sum_j P dP (tests/attn_bwd_err.py)"),
             ("e2e_idle_gap.txt", "end-to-end phases after a host-side gap: an idle GPU drops its clocks"),
             ("defer_breakdown.txt", "the slab-summing GroupNorm inside the replayed step graphs, per kernel instantiation (tools/defer_breakdown.sh)"),
             ("stamp_ring.txt", "in-kernel phases of the 256 x 320 ring tiles on the final library (tools/stamp_ring.py): entry -> first k-step 2.4-3.7 us, drain + epilogue + stores 8-12 us"),
             ("icache_cold_probe.txt", "cold instruction fetch: ~650 cycles once per launch, nothing per KiB of code (tools/probes/icache_cold.hip; a refuted hypothesis about the launches' prologue / epilogue)"),
             ("grid_barrier_probe.txt", "a dependent launch in a replayed graph (1.6 us) against a grid barrier inside one persistent launch (3.8 / 13.7 us): tools/probes/grid_barrier.hip")]
    have = set(os.listdir(dst))
    for suffix, what in known:
        if f"{tag}_{suffix}" in have:
            lines.append(f"* `{tag}_{suffix}` {what}")
    listed = {f"{tag}_{sfx}" for sfx, _ in known} | {f"{tag}_README.md"}
    for name in sorted(have):
        if name.startswith(tag + "_") and name not in listed:
            lines.append(f"* `{name}`")
    lines.append("* `graph_breakdown`, `nusc256_kernel_stats`, `nusc256_pmc_traffic`, `train_*`, `mall_share`, `vae_decode_fp16` come from "
                 "`tools/profile_more.sh` / `tools/profile_extra.sh`, last run on the build of commit 73e84cb (before the GroupNorm summed "
                 "split-K slabs: those files still show the 37 / 64 reduce launches per step)")
    lines.append("* the igemm main-loop A/Bs, split-K sweeps and the attention / adapter / feed-forward / row-chain labs of the kernels this "
                 "round left untouched are rounds 2-4's (`r02_*`, `r03_*`, `r04_*`)")
    with open(os.path.join(dst, f"{tag}_README.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
