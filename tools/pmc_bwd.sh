#!/bin/bash
# SQ counters of the backward kernels (tools/bwd_sweep.py as the workload) -> gpurun_out/pmc_bwd{A,B}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_bwdA -- python $R/tools/bwd_sweep.py --shape bsds300 --reps 2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_bwdB -- python $R/tools/bwd_sweep.py --shape bsds300 --reps 2 > /dev/null 2>&1
