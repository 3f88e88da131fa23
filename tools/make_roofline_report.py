#!/usr/bin/env python
"""profiles/r01/roofline_report.{md,json} + profiles/hbm_traffic.json from the CSVs tools/profile_bench.sh collected
(copied to profiles/r01/bench_<workload>_{kernel_stats,pmc_summary}.csv)."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "profiles", "r01")
FL = {"bsds300": 793858867200.0, "power": 92292000000.0}
ALG = {"bsds300": 8192 * 63 * (4 * (1 + 30 + 2) + 8), "power": 10000 * 6 * (4 * (1 + 30 + 2) + 8)}
TITLES = (("bsds300", "C3 BSDS300-shaped: 8192 x 63 integrals, n=100, 31-50^4-1"),
          ("power", "C2 POWER-shaped: 10000 x 6 integrals, n=100, 31-50^4-1"))
out = {}
lines = ["# Roofline report (rocprofv3, round 1) — `cc_fwd_bf16_kernel<4,2,2,EXACT,LIVE=13,PIPE>`", "",
         "Collected with `tools/profile_bench.sh <workload>` on one MI355X (kernel-trace stats, then PMC passes: SQ counters in two",
         "passes, FETCH_SIZE and WRITE_SIZE each in a pass of its own), rendered by `tools/make_roofline_report.py`.  Raw summaries:",
         "`bench_<workload>_kernel_stats.csv`, `bench_<workload>_pmc_summary.csv` next to this file.  FETCH_SIZE is doubled (gfx950",
         "tallies 128-B requests at 64 B, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs.", ""]
for w, title in TITLES:
    st = [r for r in csv.DictReader(open(os.path.join(R, f"bench_{w}_kernel_stats.csv"))) if "cc_fwd_bf16" in r["Name"]][0]
    pm = {r["counter"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(os.path.join(R, f"bench_{w}_pmc_summary.csv")))
          if "cc_fwd_bf16" in r["kernel"]}
    avg_ms = float(st["AverageNs"]) / 1e6
    cyc = pm["GRBM_GUI_ACTIVE"] / 8
    hbm = pm["FETCH_SIZE"] * 1024 * 2 + pm["WRITE_SIZE"] * 1024
    mfma_busy = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    valu = pm["SQ_INSTS_VALU"] - pm["SQ_INSTS_MFMA"]
    out[w] = dict(kernel=st["Name"], avg_launch_ms=avg_ms, calls=int(st["Calls"]), algorithmic_tflops=FL[w] / avg_ms / 1e9,
                  hbm_bytes_per_launch=hbm, algorithmic_bytes_per_launch=ALG[w], hbm_gbps=hbm / avg_ms / 1e6,
                  mfma_pipe_busy=mfma_busy, clock_ghz=cyc / avg_ms / 1e6, mfma_per_launch=pm["SQ_INSTS_MFMA"],
                  valu_per_launch=valu, lds_per_launch=pm["SQ_INSTS_LDS"], fetch_size_kb_raw=pm["FETCH_SIZE"],
                  write_size_kb_raw=pm["WRITE_SIZE"])
    tf = FL[w] / avg_ms / 1e9
    ex = pm["SQ_INSTS_MFMA"] * 16384 / avg_ms / 1e9
    lines += [f"## {title}", "", "| quantity | value |", "|---|---|",
              f"| average launch (kernel-trace, {st['Calls']} launches) | {avg_ms:.3f} ms |",
              f"| algorithmic FLOPs per launch (SURVEY 8d) | {FL[w]/1e9:.1f} GFLOP -> **{tf:.0f} TFLOP/s** |",
              f"| vs dense bf16 MFMA peak (2500 TFLOP/s) / fp32 MFMA peak (157.3) | {tf/2500:.3f} / {tf/157.3:.2f} |",
              f"| executed MFMA instructions (16x16x32 bf16) | {pm['SQ_INSTS_MFMA']/1e6:.1f} M = {pm['SQ_INSTS_MFMA']*16384/1e12:.2f} TFLOP on the pipe ({ex:.0f} TFLOP/s, {ex/2500:.2f} of peak) |",
              f"| matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles) | {100*mfma_busy:.1f} % |",
              f"| other VALU instructions / MFMA | {valu/1e6:.0f} M / {pm['SQ_INSTS_MFMA']/1e6:.0f} M = {valu/pm['SQ_INSTS_MFMA']:.2f} |",
              f"| wave time: issuing / issue-stalled / parked (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES) | {100*pm['SQ_ACTIVE_INST_ANY']/pm['SQ_WAVE_CYCLES']:.0f} % / {100*pm['SQ_WAIT_INST_ANY']/pm['SQ_WAVE_CYCLES']:.0f} % / {100*pm['SQ_WAIT_ANY']/pm['SQ_WAVE_CYCLES']:.0f} % |",
              f"| HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE) | {hbm/1e6:.1f} MB vs {ALG[w]/1e6:.1f} MB algorithmic -> {hbm/avg_ms/1e6:.0f} GB/s = {hbm/avg_ms/1e6/8000*100:.2f} % of 8 TB/s |",
              f"| shader clock during the kernel (GRBM_GUI_ACTIVE / 8 / duration) | {cyc/avg_ms/1e6:.2f} GHz |", ""]
lines += ["Reading: the path is three orders of magnitude above the HBM ridge, traffic equals the algorithmic bytes (no re-reads), and",
          "the matrix pipe is the busiest unit but waits on instruction issue: every MFMA comes with ~3 other vector instructions",
          "(activation, bf16 pieces, packing) of which about two fit in its issue shadow (DESIGN.md 4.0).  The executed-MFMA rate is 3",
          "bf16 products per algorithmic fp32 product on 64x64-padded 51x51 layers (plus the split-remainder MFMAs), hence the gap",
          "between the algorithmic and the pipe rates.", ""]
open(os.path.join(R, "roofline_report.md"), "w").write("\n".join(lines))
json.dump(out, open(os.path.join(R, "roofline_report.json"), "w"), indent=1)
method = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload {w} --steps 2 --warmup 1`; mean "
          "per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); see "
          "profiles/r01/bench_{w}_pmc_summary.csv")
json.dump({w: dict(hbm_bytes_per_launch=o["hbm_bytes_per_launch"], fetch_size_kb_raw=o["fetch_size_kb_raw"],
                   write_size_kb_raw=o["write_size_kb_raw"], kernel=o["kernel"], method=method.format(w=w),
                   algorithmic_bytes_per_launch=o["algorithmic_bytes_per_launch"]) for w, o in out.items()},
          open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print("\n".join(lines))
