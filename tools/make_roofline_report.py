#!/usr/bin/env python
"""profiles/<round>/roofline_report.{md,json} from what tools/profile_bench.sh collected under gpurun_out/prof_<tag>/.

    python tools/make_roofline_report.py r02          # copies the summaries it used into profiles/r02/ next to the report

Per tag (bsds300, power, bsds300_train ...): kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_summary.csv (two SQ
passes, mean per dispatch), bench_stats.json (the bench line of the stats pass) and, for eval tags, hbm_traffic.json
(FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md, GRBM_GUI_ACTIVE summed over the 8 XCDs).
Algorithmic FLOPs: SURVEY 8(d) per forward integral x integrals per launch; backward = 3 x that (include/umnn_cc.h)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
OUT = os.path.join(ROOT, "profiles", RND)
os.makedirs(OUT, exist_ok=True)
PEAK_BF16 = 2500.0


def flops_per_integral(cfg):
    hd, E, n = cfg["hd"], cfg["E"], cfg["n"]
    node = hd[0] + hd[-1] + sum(hd[i] * hd[i + 1] for i in range(len(hd) - 1))
    return 2.0 * ((n + 1) * node + E * hd[0])


def load(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    if not os.path.isdir(src):
        return None
    for f in ("kernel_stats.csv", "pmc_summary.csv", "bench_stats.json", "hbm_traffic.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(OUT, f"bench_{tag}_{f}"))
    stats = list(csv.DictReader(open(os.path.join(OUT, f"bench_{tag}_kernel_stats.csv"))))
    pmc = {}
    for r in csv.DictReader(open(os.path.join(OUT, f"bench_{tag}_pmc_summary.csv"))):
        pmc.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
    line = json.loads(open(os.path.join(OUT, f"bench_{tag}_bench_stats.json")).read().strip().splitlines()[-1])
    return stats, pmc, line


def kernel_entry(stats, pmc, name_part, flops, algo_bytes=None, hbm=None):
    # (several kernels may carry the name part -- since round 5 every fp16-piece forward launch is followed by the bf16 build of the
    # same kernel, queued as its overflow fallback and returning at once: take the one with the largest total time / MFMA count)
    st = max((r for r in stats if name_part in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
    pm = max((v for k, v in pmc.items() if name_part in k), key=lambda v: v.get("SQ_INSTS_MFMA", 0.0))
    avg_ms = float(st["AverageNs"]) / 1e6
    cyc = pm["GRBM_GUI_ACTIVE"] / 8
    mfma = pm["SQ_INSTS_MFMA"]
    valu = pm["SQ_INSTS_VALU"] - mfma
    e = dict(kernel=st["Name"], calls=int(st["Calls"]), avg_launch_ms=avg_ms, share_of_gpu_time_pct=float(st["Percentage"]),
             algorithmic_flops_per_launch=flops, algorithmic_tflops=flops / avg_ms / 1e9,
             frac_of_bf16_peak=flops / avg_ms / 1e9 / PEAK_BF16,
             mfma_per_launch=mfma, executed_tflops=mfma * 16384 / avg_ms / 1e9, executed_frac_of_bf16_peak=mfma * 16384 / avg_ms / 1e9 / PEAK_BF16,
             mfma_pipe_busy=pm["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, other_valu_per_mfma=valu / mfma, lds_per_mfma=pm["SQ_INSTS_LDS"] / mfma,
             wave_issuing=pm["SQ_ACTIVE_INST_ANY"] / pm["SQ_WAVE_CYCLES"], wave_issue_stalled=pm["SQ_WAIT_INST_ANY"] / pm["SQ_WAVE_CYCLES"],
             wave_parked=pm["SQ_WAIT_ANY"] / pm["SQ_WAVE_CYCLES"], clock_ghz=cyc / avg_ms / 1e6)
    if hbm is not None:
        e.update(hbm_bytes_per_launch=hbm, algorithmic_bytes_per_launch=algo_bytes, hbm_gbps=hbm / avg_ms / 1e6)
    return e


def table(title, e):
    rows = [f"### {title}", "", "| quantity | value |", "|---|---|",
            f"| kernel | `{e['kernel'][:90]}` |",
            f"| average launch (kernel-trace, {e['calls']} launches) | **{e['avg_launch_ms']:.3f} ms** ({e['share_of_gpu_time_pct']:.1f} % of the GPU time of the run) |",
            f"| algorithmic FLOPs per launch | {e['algorithmic_flops_per_launch']/1e9:.1f} GFLOP -> **{e['algorithmic_tflops']:.0f} TFLOP/s** = {e['frac_of_bf16_peak']:.3f} of the dense bf16 MFMA peak (2500) |",
            f"| executed MFMA (16x16x32-class, 16384 FLOP each) | {e['mfma_per_launch']/1e6:.1f} M -> {e['executed_tflops']:.0f} TFLOP/s on the pipe = {e['executed_frac_of_bf16_peak']:.2f} of peak |",
            f"| matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles) | **{100*e['mfma_pipe_busy']:.1f} %** |",
            f"| other VALU / MFMA, LDS instructions / MFMA | {e['other_valu_per_mfma']:.2f}, {e['lds_per_mfma']:.2f} |",
            f"| wave time issuing / issue-stalled / parked | {100*e['wave_issuing']:.0f} % / {100*e['wave_issue_stalled']:.0f} % / {100*e['wave_parked']:.0f} % |",
            f"| shader clock during the kernel | {e['clock_ghz']:.2f} GHz |"]
    if "hbm_bytes_per_launch" in e:
        rows.append(f"| HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE) | {e['hbm_bytes_per_launch']/1e6:.1f} MB vs {e['algorithmic_bytes_per_launch']/1e6:.1f} MB "
                    f"algorithmic -> {e['hbm_gbps']:.0f} GB/s = {e['hbm_gbps']/80:.2f} % of 8 TB/s |")
    return rows + [""]


report, lines = {}, [f"# Roofline report (rocprofv3, round {RND[1:]}) -- kernels at HEAD", "",
                     "Collected with `tools/profile_bench.sh <workload> [--mode train]` on one MI355X, rendered by `tools/make_roofline_report.py`;",
                     "the CSV / JSON summaries it read are the `bench_<tag>_*` files next to this report.", ""]
for tag, title in (("bsds300", "C3 BSDS300-shaped eval (8192 x 63 integrals per launch, n=100, 31-50^4-1)"),
                   ("power", "C2 POWER-shaped eval (10000 x 6 integrals per launch)"),
                   ("toy", "C1 2-moons eval (4096 x 2 integrals per launch, n=50, 11-100^4-1)"),
                   ("vae", "C4 VAE prior flow eval (1024 x 64 integrals per launch, n=50, cond 320)"),
                   ("mnist", "MNISTExperiment-shaped eval (100 x 784 integrals per launch, n=50, 31-100-50^4-1; the d=784 shape BASELINE config 5 quotes)"),
                   ("bsds300_train", "C3 training step (forward + HIP backward + Adam)"),
                   ("mnist_train", "MNISTExperiment-shaped training step (d=784, 31-100-50-50-50-50-1, batch 100; three-stage backward)")):
    got = load(tag)
    if got is None:
        continue
    stats, pmc, line = got
    cfg = bench.WORKLOADS[tag.replace("_train", "")]
    fl = flops_per_integral(cfg) * cfg["rows"] * cfg["d"]
    lines += [f"## {title}", "", f"bench line of the stats pass: {line['value']:.4g} {line['unit']}, {line['ms_per_step']:.3f} ms/step.", ""]
    report[tag] = {"bench": {k: line[k] for k in ("value", "unit", "ms_per_step")}}
    hbm = algo = None
    tr = os.path.join(OUT, f"bench_{tag}_hbm_traffic.json")
    if os.path.exists(tr):
        rec = json.load(open(tr)).get(tag, {})
        hbm, algo = rec.get("hbm_bytes_per_launch"), rec.get("algorithmic_bytes_per_launch")
    e = kernel_entry(stats, pmc, "cc_fwd_bf16_kernel", fl, algo, hbm)
    report[tag]["forward"] = e
    lines += table("forward quadrature kernel", e)
    if tag.endswith("_train"):
        if tag.startswith("mnist"):     # three kernels per chunk share the backward's work: report each with its own counters
            for part, ttl in ((next((k for k in ("cc_front_fwd16_kernel",) if any(k in r["Name"] for r in stats)), "cc_front_fwd_kernel"),
                               "stage A: front forward (a1, z2 -> HBM; round 4: on fp16 pieces, two workgroups per CU -- the cc_front_fwd_kernel "
                               "launches next to it are the queued overflow fallback returning at once)"),
                              (next((k for k in ("cc_bwd_ws16_kernel", "cc_bwd_ws_kernel") if any(k in r["Name"] for r in stats)), "cc_bwd_bf16_kernel"),
                               "stage B: flagship kernel on the net from hidden layer 2 on (FRONT; the workgroup pipeline, round 4: on fp16 pieces -- "
                               "the cc_bwd_ws_kernel launches next to it are the queued overflow fallback returning at once)"),
                              (next((k for k in ("cc_front_bwd16_kernel", "cc_front_bwd2_kernel") if any(k in r["Name"] for r in stats)), "cc_front_bwd_kernel"),
                               "stage C: front backward (dG1, delta_1; round 6: two waves per tile of integrals, dG1 on 32x32x16 through LDS-transposed "
                               "operands, on fp16 pieces -- cc_front_bwd16_kernel; the cc_front_bwd2_kernel launches next to it are its queued bf16 "
                               "fallback returning at once)")):
                e = kernel_entry(stats, pmc, part, fl)
                e["algorithmic_flops_per_launch"] = None
                report[tag][part] = e
                rows = table(ttl, dict(e, algorithmic_flops_per_launch=0.0, algorithmic_tflops=0.0, frac_of_bf16_peak=0.0))
                lines += [r for r in rows if "algorithmic FLOPs" not in r]
        else:
            # (round 3: the software-pipelined loop is its own kernel symbol; older profiles have cc_bwd_bf16_kernel)
            # (round 4: cc_bwd_ws16_kernel, with the bf16 cc_bwd_ws_kernel queued behind it as the overflow fallback that returns at once)
            name = next((k for k in ("cc_bwd_ws16_kernel", "cc_bwd_ws_kernel", "cc_bwd_swp_kernel") if any(k in r["Name"] for r in stats)), "cc_bwd_bf16_kernel")
            e = kernel_entry(stats, pmc, name, 3 * fl)
            report[tag]["backward"] = e
            lines += table("backward quadrature kernel (algorithmic FLOPs = 3 x forward: two gradient GEMMs per forward GEMM + the recompute)", e)
        lines += ["Top kernels of the training step (kernel-trace):", "", "| kernel | calls | avg | % |", "|---|---|---|---|"]
        for r in stats[:8]:
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} ms | {float(r['Percentage']):.1f} |")
        lines.append("")
open(os.path.join(OUT, "roofline_report.md"), "w").write("\n".join(lines))
json.dump(report, open(os.path.join(OUT, "roofline_report.json"), "w"), indent=1)
print("\n".join(lines))
