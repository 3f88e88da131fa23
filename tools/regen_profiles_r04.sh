#!/bin/bash
# Everything under profiles/r04/ in one go (on the GPU box: `gpurun -- 'bash tools/regen_profiles_r04.sh'`), then locally
# `bash tools/regen_profiles_r04.sh collect` copies the summaries from gpurun_out/ into profiles/r04/ and FAILS if
# profiles/hbm_traffic.json was not measured on HEAD's forward-kernel sources.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
if [ "$1" = collect ]; then
  set -e
  D=profiles/r04; mkdir -p $D
  for t in bsds300 bsds300_train power power_train mnist_train vae; do
    for f in kernel_stats.csv pmc_summary.csv bench_stats.json hbm_traffic.json; do
      [ -f gpurun_out/prof_$t/$f ] && cp gpurun_out/prof_$t/$f $D/bench_${t}_$f
    done
  done
  cp gpurun_out/bench_lines.jsonl $D/bench_lines.jsonl
  cp gpurun_out/ws16_role_timing.txt $D/bwd_ws16_role_timing.txt
  cp gpurun_out/pmc_bwd_ws1/summary.csv $D/bwd_ws16_pmc.csv
  [ -f gpurun_out/pmc_bwd_ws16off/summary.csv ] && cp gpurun_out/pmc_bwd_ws16off/summary.csv $D/bwd_ws_bf16_pmc.csv
  cp gpurun_out/kink_rows.txt $D/kink_rows.txt; cp gpurun_out/bwd_truth64.txt $D/bwd_truth64.txt
  cp gpurun_out/f16_split.txt $D/f16_split.txt
  cp gpurun_out/prof_bsds300/hbm_traffic.json profiles/hbm_traffic.json
  python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
rec = json.load(open("profiles/hbm_traffic.json"))
bad = [w for w, r in rec.items() if r.get("source_sha256") != bench.kernel_source_hash()]
assert "bsds300" in rec and "bsds300" not in bad, f"profiles/hbm_traffic.json is stale for bsds300 (HEAD hashes to {bench.kernel_source_hash()})"
print("hbm_traffic.json fresh for", [w for w in rec if w not in bad], "stale:", bad)
PY
  python tools/make_roofline_report.py r04
  exit 0
fi
set -x
bash tools/profile_bench.sh bsds300 > /dev/null 2>&1
bash tools/profile_bench.sh bsds300 --mode train > /dev/null 2>&1
bash tools/profile_bench.sh power > /dev/null 2>&1
bash tools/profile_bench.sh power --mode train > /dev/null 2>&1
bash tools/profile_bench.sh mnist --mode train > /dev/null 2>&1
bash tools/profile_bench.sh vae > /dev/null 2>&1
bash tools/bench_lines.sh > gpurun_out/bench_lines.txt 2>&1
[ -f umnn_amd/libumnn_cc_w16timing.so ] && UMNN_CC_LIB=$PWD/umnn_amd/libumnn_cc_w16timing.so timeout 200 python tools/bwd_sweep.py --shape bsds300 --reps 2 2>&1 | grep "WS16_TIMING\|backward" | awk '!seen[$0]++' > gpurun_out/ws16_role_timing.txt
bash tools/pmc_bwd_ws.sh 1 > gpurun_out/pmc_bwd_ws.txt 2>&1
timeout 300 python tools/kink_rows.py 2>&1 | grep -v amdgpu.ids > gpurun_out/kink_rows.txt
timeout 300 python tools/bwd_truth64.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bwd_truth64.txt
./tools/ubench/f16_split > gpurun_out/f16_split.txt 2>&1
cat gpurun_out/bench_lines.txt | tail -16; cat gpurun_out/ws16_role_timing.txt
