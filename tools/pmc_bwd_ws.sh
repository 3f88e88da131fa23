#!/bin/bash
# SQ counters of the one-pass backward kernels at C3: weight-stationary workgroup pipeline (UMNN_BWD_WS=1) vs the software-pipelined
# loop (UMNN_BWD_WS=0), tools/bwd_sweep.py as the workload -> gpurun_out/pmc_bwd_ws{0,1}/summary.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for S in ${1:-0 1}; do
  O=$R/gpurun_out/pmc_bwd_ws$S; rm -rf $O; mkdir -p $O
  UMNN_BWD_WS=$S timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python $R/tools/bwd_sweep.py --shape bsds300 --reps 2 > /dev/null 2>&1
  UMNN_BWD_WS=$S timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python $R/tools/bwd_sweep.py --shape bsds300 --reps 2 > /dev/null 2>&1
  UMNN_BWD_WS=$S timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_ANY --output-format csv -d $O/pmc3 -- python $R/tools/bwd_sweep.py --shape bsds300 --reps 2 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O | grep -i "cc_bwd_bf16\|cc_bwd_swp\|cc_bwd_ws\|kernel,counter" > $O/summary.csv
  find $O -name "*_counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
  cat $O/summary.csv
done
