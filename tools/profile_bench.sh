#!/bin/bash
# rocprofv3 passes behind profiles/: kernel-trace stats + two PMC passes for `bench.py --workload $1` (run on the GPU box).
#   bash tools/profile_bench.sh bsds300|power|toy [extra bench args]   -> gpurun_out/prof_<workload>/{stats,pmc1,pmc2,pmc_FETCH_SIZE,pmc_WRITE_SIZE}; every pass under its own timeout
W=${1:-bsds300}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_$W; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --workload $W --no-cpu-baseline --steps 10 --warmup 3 "$@" > $O/bench_stats.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python $R/bench.py --workload $W --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc2 -- python $R/bench.py --workload $W --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2>&1
# memory-traffic counters: one counter per pass (together, or mixed with SQ counters, the run never finished here)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --workload $W --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $O > $O/pmc_summary.csv
