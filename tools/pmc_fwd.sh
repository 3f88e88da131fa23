R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for p in 0 1; do
  UMNN_FWD_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcA_$p -- python $R/tools/fwd_sweep.py --shape bsds300 --reps 3 > /dev/null 2>&1
  UMNN_FWD_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmcB_$p -- python $R/tools/fwd_sweep.py --shape bsds300 --reps 3 > /dev/null 2>&1
done
