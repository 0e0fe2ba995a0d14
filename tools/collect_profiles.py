#!/usr/bin/env python3
"""Copy the summaries of one `tools/profile_round.sh <tag>` call from gpurun_out/<tag>/ into profiles/ and regenerate
profiles/<tag>_README.md from them (kernel table from the rocprofv3 stats, dominant-kernel cross-check against the
bench line's own HIP-event brackets, PMC traffic).

    python tools/collect_profiles.py r01
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))
    cp("kernel_stats.csv", "unet512_b16_bf16_kernel_stats.csv")
    cp("bench.json", "bench_unet512_b16_bf16.json")
    cp("launches.tsv", "launches_one_step.tsv")
    cp("pmc_traffic.json", "pmc_traffic.json")
    if os.path.exists(os.path.join(src, "pp_phase_stamps.txt")):
        cp("pp_phase_stamps.txt", "pp_phase_stamps.txt")
    bench = json.load(open(os.path.join(src, "bench.json")))
    pmc = json.load(open(os.path.join(src, "pmc_traffic.json")))
    rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
    roof = bench["roofline"]
    kname = roof["kernel"]
    dom = [r for r in rows if kname in r["Name"]]
    calls = sum(int(r["Calls"]) for r in dom)
    tot_ns = sum(float(r["TotalDurationNs"]) for r in dom)
    lines = [f"# Round 1 profiles (MI355X, 1 GPU) -- tag {tag}", "",
             f"All files come from ONE GPU-box call of `bash tools/profile_round.sh {tag}` (bench line, launch dump, rocprofv3",
             "`--kernel-trace --stats`, and one `--pmc` pass per counter of the same bench command), copied here by",
             "`tools/collect_profiles.py`.", "",
             f"Workload: {bench['config']['workload']}, {bench['dtype']}; 7 steps in the trace",
             "(`python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-e2e`).", "",
             f"Bench line of the same box: {bench['value']:.1f} {bench['unit']}, {bench['ms_per_step']:.2f} ms per step, "
             f"model rate {bench['model_tflops']:.0f} TFLOP/s ({100 * bench['model_frac_of_peak']:.1f} % of the bf16 MFMA peak), "
             f"{bench.get('objects_per_s', 0):.2f} inpainted objects/s end to end, CPU baseline "
             f"{bench['cpu_baseline']['value']:.3f} {bench['cpu_baseline']['unit']} on {bench['cpu_baseline']['cores']} threads.", "",
             "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:18]:
        lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                     f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines += ["",
              f"Dominant kernel = `{kname}<...>` ({len(dom)} template instances in the table): {calls} calls, {tot_ns / 1e6:.1f} ms, average "
              f"{tot_ns / calls / 1e3:.1f} us under rocprofv3 vs `roofline.avg_launch_us` = {roof['avg_launch_us']} us from bench.py's own HIP-event "
              "brackets on the same box -- same kernel, same command.  `roofline.achieved` = "
              f"{roof['achieved']} TFLOP/s = {roof['gflop_per_launch']} GFLOP (algorithmic 2*M*N*K) / that launch time; frac {roof['frac']}.",
              "`__amd_rocclr_copyBuffer` rows are the one-off weight upload / packing, not per-step work.", ""]
    rec = pmc.get(kname)
    if rec:
        lines += [f"HBM traffic (`{tag}_pmc_traffic.json`, FETCH_SIZE x 2 + WRITE_SIZE, separate `--pmc` passes): "
                  f"{rec['hbm_bytes_per_launch_corrected'] / 1e6:.1f} MB per `{kname}` launch against "
                  f"{roof['algorithmic_mb_per_launch']} MB algorithmic (`roofline.algorithmic_mb_per_launch`): the nine taps of a 3x3 window "
                  "re-fetch their pixels.  Removing those re-fetches does not make the loop faster (DESIGN.md section 6).", ""]
    lines += ["Files:",
              f"* `{tag}_unet512_b16_bf16_kernel_stats.csv` rocprofv3 stats; `{tag}_bench_unet512_b16_bf16.json` bench line of the same box",
              f"* `{tag}_launches_one_step.tsv` kind, GFLOP, us, algorithmic MB, kernel variant and shape of every launch of one step",
              f"* `{tag}_pmc_traffic.json` per-kernel-family FETCH_SIZE / WRITE_SIZE (tools/pmc_summary.py)",
              f"* `{tag}_pp_phase_stamps.txt` per-phase `s_memtime` stamps of the ping-pong kernel (tools/stamp_pp.py, -DMOBI_STAMP=3):",
              "  cycles per k-tile at the LOAD barrier, in the LOAD phase, at the MATRIX barrier, in the MATRIX phase, in the counted",
              "  vmcnt wait (late / early half)",
              f"* `{tag}_pp_ablation.txt` the k loop with parts removed (tools/diag_ingest.sh on the lockstep kernel, the same ablations of the ping-pong / halo kernels (the halo kernel and its script were removed later) and",
              "  tools/ab_flags.sh on the ping-pong / halo kernels): no activation DMA / no weight DMA / no DMA / no MFMA",
              f"* `{tag}_error_table.txt` measured rel-L2 vs the fp32 oracle per storage type (tests/error_table.py)",
              f"* `{tag}_igemm_phase_stamps.txt`, `{tag}_splitk_sweep.txt` earlier sessions' stamps of the lockstep kernel and the split-K plan sweep",
              f"* `{tag}_pp_sq_counters.txt` SQ counters of the ping-pong igemm on one 3x3 launch (tools/pmc_kernel.sh); `{tag}_attention_ablation.txt` holds the attention kernel's",
              f"* `{tag}_other_configs.txt` fp16 / CFG / mobi_nusc_256 / 2-rank-gloo bench lines of the final build"]
    with open(os.path.join(dst, f"{tag}_README.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
