#!/usr/bin/env python
"""HBM bytes per launch of the dominant forward kernel of a bench.py workload, from rocprofv3 PMC counters.

Two passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC has 4 slots, they cost 3 + 2), each a child run of
`bench.py --workload W --steps 2 --warmup 1 --no-cpu-baseline --no-extras` under `rocprofv3 --kernel-trace --pmc <counter>`
with its own timeout.  Corrections per MI355X_MICROARCH.md (HBM): FETCH_SIZE is reported in KiB and, on gfx950, tallies the
128-byte requests of wide coalesced reads at 64 bytes -> doubled; WRITE_SIZE in KiB as is.  Mean per dispatch of the kernel
with the largest total.

    python tools/measure_traffic.py bsds300 [--update]      # --update rewrites profiles/hbm_traffic.json (with the source hash)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pass(counter, workload, extra, outdir):
    cmd = ["timeout", "240", "rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--no-telemetry"] + list(extra)
    env = dict(os.environ, TMPDIR="/tmp", UMNN_BENCH_CHILD="1")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"rocprofv3 --pmc {counter} failed ({r.returncode}): {r.stderr[-400:]}")
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    if not acc:
        raise RuntimeError(f"no {counter} rows in {outdir}")
    return acc


def measure(workload, extra=()):
    import bench
    tmp = tempfile.mkdtemp(prefix="umnn_traffic_", dir="/tmp")
    try:
        fetch = _pass("FETCH_SIZE", workload, extra, os.path.join(tmp, "fetch"))
        write = _pass("WRITE_SIZE", workload, extra, os.path.join(tmp, "write"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    cands = [k for k in fetch if "cc_fwd" in k] or list(fetch)
    kernel = max(cands, key=lambda k: sum(fetch[k]))
    f_kb = sum(fetch[kernel]) / len(fetch[kernel])
    w_kb = sum(write[kernel]) / len(write[kernel]) if kernel in write else 0.0
    cfg = bench.WORKLOADS[workload]
    return {"hbm_bytes_per_launch": f_kb * 1024 * 2 + w_kb * 1024, "fetch_size_kb_raw": f_kb, "write_size_kb_raw": w_kb,
            "kernel": kernel, "dispatches": len(fetch[kernel]), "source_sha256": bench.kernel_source_hash(),
            "algorithmic_bytes_per_launch": cfg["rows"] * cfg["d"] * (4 * (1 + cfg["E"] + 2) + 8),
            "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload %s --steps 2 "
                      "--warmup 1 --no-cpu-baseline --no-extras`; mean per dispatch; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                      "(gfx950 tallies 128-B requests at 64 B)" % workload}


if __name__ == "__main__":
    w = sys.argv[1] if len(sys.argv) > 1 else "bsds300"
    rec = measure(w)
    print(json.dumps(rec, indent=1))
    if "--update" in sys.argv:
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[w] = rec
        json.dump(cur, open(path, "w"), indent=1)
