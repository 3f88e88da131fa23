#!/bin/bash
# Same-box A/B of two builds of the library (fresh boxes differ by a few per cent): alternates `bench.py "$@"` with
# UMNN_CC_LIB=<A> and <B>, three rounds, one line per run.
#   bash tools/ab_bench.sh umnn_amd/libumnn_cc_base.so umnn_amd/libumnn_cc.so [bench args]
A=$(realpath $1); B=$(realpath $2); shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2 3; do
  for L in $A $B; do
    UMNN_CC_LIB=$L timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; t=d.get('train_kernels'); print('$(basename $L)', round(d['value']), round(d['ms_per_step'],3), round(r.get('avg_launch_ms',0),4), (round(t['backward_main']['avg_launch_ms'],3) if t else ''))"
  done
done
