#!/bin/bash
# Everything under profiles/r06/ in one go (on the GPU box: `gpurun -- 'bash tools/regen_profiles_r06.sh'`), then locally
# `bash tools/regen_profiles_r06.sh collect` copies the summaries from gpurun_out/ into profiles/r06/, MERGES the per-workload HBM
# traffic records into profiles/hbm_traffic.json and FAILS if ANY key of that file was not measured on HEAD's forward-kernel sources
# (the `power` key had stayed round 2's because only the bsds300 record was copied back; VERDICT r05 item 5: `mnist`
# eval is profiled too, so the file has its key).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
if [ "$1" = collect ]; then
  set -e
  D=profiles/r06; mkdir -p $D
  for t in bsds300 bsds300_train power power_train mnist mnist_train vae toy; do
    for f in kernel_stats.csv pmc_summary.csv bench_stats.json hbm_traffic.json; do
      [ -f gpurun_out/prof_$t/$f ] && cp gpurun_out/prof_$t/$f $D/bench_${t}_$f
    done
  done
  cp gpurun_out/bench_lines.jsonl $D/bench_lines.jsonl
  [ -f gpurun_out/bwd_truth64_c3.txt ] && cp gpurun_out/bwd_truth64_c3.txt gpurun_out/bwd_truth64_mnist.txt $D/
  [ -f gpurun_out/pmc_bwd_ws1/summary.csv ] && cp gpurun_out/pmc_bwd_ws1/summary.csv $D/bwd_ws16_pmc.csv
  [ -f gpurun_out/ops_power.txt ] && cp gpurun_out/ops_power.txt $D/train_step_ops_power.txt
  [ -f gpurun_out/fwd_energy.json ] && cp gpurun_out/fwd_energy.json $D/fwd_energy.json
  python - <<'PY'
import glob, json, sys
sys.path.insert(0, ".")
import bench
merged = {}
for f in sorted(glob.glob("gpurun_out/prof_*/hbm_traffic.json")):
    w = f.split("prof_")[1].split("/")[0]
    rec = json.load(open(f))
    if w in rec:
        merged[w] = rec[w]
json.dump(merged, open("profiles/hbm_traffic.json", "w"), indent=1)
bad = [w for w, r in merged.items() if r.get("source_sha256") != bench.kernel_source_hash()]
assert "bsds300" in merged and "mnist" in merged, "no bsds300 / mnist traffic record"
assert not bad, f"profiles/hbm_traffic.json is stale for {bad} (HEAD hashes to {bench.kernel_source_hash()})"
print("hbm_traffic.json fresh for", sorted(merged))
PY
  python tools/make_roofline_report.py r06
  exit 0
fi
set -x
bash tools/profile_bench.sh bsds300 > /dev/null 2>&1
bash tools/profile_bench.sh bsds300 --mode train > /dev/null 2>&1
bash tools/profile_bench.sh power > /dev/null 2>&1
bash tools/profile_bench.sh power --mode train > /dev/null 2>&1
bash tools/profile_bench.sh mnist > /dev/null 2>&1
bash tools/profile_bench.sh mnist --mode train > /dev/null 2>&1
bash tools/profile_bench.sh vae > /dev/null 2>&1
bash tools/profile_bench.sh toy > /dev/null 2>&1
bash tools/bench_lines.sh > gpurun_out/bench_lines.txt 2>&1
bash tools/pmc_bwd_ws.sh 1 > gpurun_out/pmc_bwd_ws.txt 2>&1
timeout 300 python tools/bwd_truth64_sizes.py gpurun_out 2>&1 | grep -v amdgpu.ids > gpurun_out/bwd_truth64.log
timeout 200 python tools/train_step_ops.py power 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/ops_power.txt
cat gpurun_out/bench_lines.txt | tail -16
