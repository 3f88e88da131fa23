#!/usr/bin/env python
"""How many ROWS of the per-row backward outputs (d_h; d_x with a g_fx cotangent) differ between two evaluations of the same
launch by more than a threshold -- at the benchmarked size every pair of fp32-level evaluations differs in some rows, because a
LeakyReLU kink decision inside rounding noise moves its row discontinuously.  Pairs: every kernel against the six-term bf16 loop."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd  # noqa: E402
from umnn_amd import _lib, integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
B, d, E, n = 8192, 63, 30, 100
torch.manual_seed(3)
net = umnn_amd.IntegrandNetwork(d, 1 + E, [50] * 4, 1).to(dev)
spec = mlp_spec(net)
x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
gg, gf = torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
outs = {}
for key, ws, ws16, prec in (("swp", 0, 0, "bf16x3"), ("ws", 1, 0, "bf16x3"), ("ws16", 1, 2, "bf16x3"), ("fp32", 0, 0, "fp32")):
    _lib.set_backward_precision(prec)
    with _lib.options(bwd_ws=ws, bwd_ws16=ws16):
        outs[key] = [o.cpu().numpy() for o in I.hip_backward(spec, None, x, h, gg, gf, n)[1:]]
    _lib.set_backward_precision("bf16x3")
for key in ("ws", "fp32", "ws16"):
    for i, nm in enumerate(("dx", "dh", "dtheta")):
        a_, c_ = outs["swp"][i], outs[key][i]
        if nm == "dtheta":
            print(f"{key:5s} vs swp  dtheta  max err / max {np.abs(c_ - a_).max() / np.abs(a_).max():.2e}")
            continue
        row = np.abs(c_ - a_).max(axis=1) / np.abs(a_).max()
        print(f"{key:5s} vs swp  {nm:6s} rows > 1e-5: {(row > 1e-5).sum():4d}  > 1e-4: {(row > 1e-4).sum():4d}  > 1e-3: {(row > 1e-3).sum():4d}  "
              f"max {row.max():.2e}  median {np.median(row):.1e}", flush=True)
