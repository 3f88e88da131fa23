#!/bin/bash
# rocprofv3 passes behind profiles/rNN/: kernel-trace stats + PMC passes for `bench.py --workload $1 [--mode train]`
# (run on the GPU box).  Every pass has its own timeout; SQ counters in two passes; FETCH_SIZE / WRITE_SIZE one per pass.
#   bash tools/profile_bench.sh bsds300 [--mode train ...]  -> gpurun_out/prof_<tag>/{stats,pmc1,pmc2,bench_stats.json,pmc_summary.csv}
#   and (eval mode) profiles/hbm_traffic.json refreshed for the CURRENT kernel sources.
W=${1:-bsds300}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$W; case " $* " in *" train "*) TAG=${W}_train;; esac
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --workload $W --no-cpu-baseline --no-extras --steps 10 --warmup 3 "$@" > $O/bench_stats.json 2>$O/bench_stats.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python $R/bench.py --workload $W --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc2 -- python $R/bench.py --workload $W --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > /dev/null 2>&1
python $R/tools/pmc_summary.py $O > $O/pmc_summary.csv
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
if [ "$TAG" = "$W" ]; then
  (cd $R && timeout 600 python tools/measure_traffic.py $W --update > $O/traffic.json 2>$O/traffic.err; cp profiles/hbm_traffic.json $O/hbm_traffic.json)
fi
# keep the merge-back small: the raw per-dispatch CSVs are large
find $O -name "*_counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
ls $O
