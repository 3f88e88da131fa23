#!/usr/bin/env python
"""Run the bf16 backward on fixed seeded inputs and save every output (A/B of two library builds: UMNN_CC_LIB=... python
tools/bwd_dump.py out.pt ; python tools/bwd_dump.py --compare a.pt b.pt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--compare":
    A, B = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in A:
        for i, (u, v) in enumerate(zip(A[k], B[k])):
            d = float((u - v).abs().max()) / max(float(u.abs().max()), 1e-30)
            print(k, ("dx0", "dx", "dh", "dtheta")[i], "bit-identical" if torch.equal(u, v) else f"max diff / max {d:.3e}")
    sys.exit(0)

import umnn_amd  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for (B, d, E, hid, n) in [(257, 63, 30, [50] * 4, 100), (64, 5, 8, [50] * 2, 20), (5, 3, 4, [48, 60, 36], 7), (100, 6, 30, [50] * 3, 50)]:
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.7)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg, gf = torch.randn(B, E * d, device=dev) * 3, torch.randn(B, d, device=dev), torch.randn(B, d, device=dev)
    out[str((B, d, hid))] = [t.cpu() for t in I.hip_backward(mlp_spec(net), x0, x, h, gg, gf, n)]
torch.save(out, sys.argv[1])
print("saved", sys.argv[1])
