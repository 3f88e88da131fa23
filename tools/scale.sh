#!/bin/bash
# Scaling sweep on one node: bench.py at 1, 2, 4, 8 GPUs under torch.distributed.run (one rank per GPU, RCCL), one JSON
# line per N into gpurun_out/scale_N.json.  Each line carries ranks_seen / dist.backend / per-rank device + PCI bus id, so
# the first contact with an 8-GPU node shows at a glance that RCCL saw N distinct devices.
#   bash tools/scale.sh [extra bench.py args, e.g. --mode train]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
NG=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && { echo "only $NG GPU(s) visible: stopping before N=$N"; break; }
  if [ "$N" -eq 1 ]; then
    timeout 900 python $R/bench.py --gpus 1 --steps 10 --warmup 3 "$@" | tee $R/gpurun_out/scale_$N.json
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      $R/bench.py --gpus $N --steps 10 --warmup 3 "$@" | tee $R/gpurun_out/scale_$N.json
  fi
done
