#!/usr/bin/env python
"""Which ATen ops / kernels a training step launches, and from where: torch.profiler over bench.py's own train step.
    python tools/train_step_ops.py [workload] [rows]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "power"
cfg = dict(bench.WORKLOADS[w])
if len(sys.argv) > 2:
    cfg["rows"] = int(sys.argv[2])
dev = torch.device("cuda:0")
model = bench.build_model(cfg, dev)
model.train()
x, ctx = bench.make_inputs(cfg, cfg["rows"], dev, 1000)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
step = bench.make_train_step(model, opt, x, ctx, 1)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.device_time_total > 0 or "memset" in e.key.lower() or "memcpy" in e.key.lower()]
rows.sort(key=lambda e: -e.count)
print(f"{'op':60s} {'calls/step':>10s} {'dev us/step':>12s}  stack")
for e in rows[:70]:
    st = " <- ".join(s.split("/")[-1] for s in e.stack[:4] if "torch/" not in s or "optim" in s or "utils" in s)
    print(f"{e.key[:60]:60s} {e.count / 3:10.1f} {e.device_time_total / 3:12.1f}  {st[:200]}")
