set -x
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh bsds300 > /dev/null 2>&1
bash tools/profile_bench.sh bsds300 --mode train > /dev/null 2>&1
bash tools/profile_bench.sh power > /dev/null 2>&1
bash tools/profile_bench.sh power --mode train > /dev/null 2>&1
bash tools/bench_lines.sh > gpurun_out/bench_lines.txt 2>&1
UMNN_CC_LIB=$PWD/umnn_amd/libumnn_cc_wstiming.so timeout 200 python tools/bwd_sweep.py --shape bsds300 --reps 2 2>&1 | grep "WS_TIMING\|backward" | awk '!seen[$0]++' > gpurun_out/ws_role_timing.txt
bash tools/pmc_bwd_ws.sh > gpurun_out/pmc_bwd_ws.txt 2>&1
cat gpurun_out/bench_lines.txt | tail -16; cat gpurun_out/ws_role_timing.txt
