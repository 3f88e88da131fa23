#!/bin/bash
# One bench.py line per workload and mode -> gpurun_out/bench_lines.jsonl (copied to profiles/rNN/bench_lines.jsonl)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bench_lines.jsonl; : > $O
run() { timeout 400 python $R/bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-20} --warmup ${WARMUP:-5} "$@" 2>/dev/null | tail -1 >> $O; }
fast() { STEPS=200 WARMUP=40 run "$@"; }          # the sub-3-ms steps: 200 timed steps
run --workload bsds300
run --workload bsds300 --precision fp32
run --workload bsds300 --precision bf16x3
run --workload bsds300 --embedding bf16
fast --workload power
fast --workload toy
fast --workload toy --graph
fast --workload vae
fast --workload vae --graph
fast --workload vae --embedding bf16
fast --workload mnist
fast --workload mnist --embedding bf16
run --workload bsds300 --mode train
run --workload power --mode train
run --workload vae --mode train
run --workload mnist --mode train
run --workload vae --mode train --embedding bf16
run --workload mnist --mode train --embedding bf16
run --workload bsds300 --mode train --graph
fast --workload power --mode train --rows 100 --graph
python - <<PY
import json
for l in open("$O"):
    d = json.loads(l); r = d.get("roofline") or {}
    print(d["config"]["workload"][:34].ljust(34), d["metric"][9:22], d["config"].get("embedding_storage"), d["dtype"][:10].ljust(10), round(d["value"]), "per s", round(d["ms_per_step"], 3), "ms  kernel", round(r.get("avg_launch_ms", 0), 3), d["config"].get("graph", "")[:20])
PY
