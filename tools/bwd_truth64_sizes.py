#!/usr/bin/env python
"""profiles/r06/bwd_truth64_c3.txt / _mnist.txt: every backward route at the two benchmarked training shapes against the float64
evaluator of the reference algorithm (tests/_truth64.py), beside a float32 run of the reference's own materialised algorithm.
    python tools/bwd_truth64_sizes.py [outdir]      (GPU box; ~1 min)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_round5 as R5  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
for tag, args, routes in (("c3", dict(B=8192, d=63, hid=[50] * 4, n=100, seed=3, wscale=1.0, gfx_scale=1.0, chunk=128), R5.C3_ROUTES),
                          ("mnist", dict(B=100, d=784, hid=[100, 50, 50, 50, 50], n=50, seed=0, wscale=1.5, gfx_scale=0.1, chunk=4), R5.MNIST_ROUTES)):
    rep, kernels = R5.backward_truth_report(dev, routes=routes, **args)
    with open(os.path.join(out, f"bwd_truth64_{tag}.txt"), "w") as f:
        f.write(f"# {args}\n# max |out - truth64| / max |truth64| per output tensor; truth = tests/_truth64.backward_reference in float64\n")
        for key, r in rep.items():
            f.write(f"{key:62s} {kernels.get(key, ''):40s} " + " ".join(f"{k} {v:.2e}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items()) + "\n")
    print(tag, json.dumps(rep), flush=True)
