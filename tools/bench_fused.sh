#!/bin/bash
# bench.py lines with the fused conditioner kernel on / off (UMNN_MADE_FUSED), one line per workload and mode
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for W in ${1:-toy power vae bsds300}; do for F in 0 1; do
  UMNN_MADE_FUSED=$F timeout 300 python $R/bench.py --workload $W --no-cpu-baseline --no-extras --no-telemetry --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W fused=$F', round(d['value']), 'evals/s', round(d['ms_per_step'],4), 'ms/step  kernel', round(d['roofline']['avg_launch_ms'],4))"
done; done
