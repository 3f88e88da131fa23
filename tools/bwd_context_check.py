#!/usr/bin/env python
"""Does the backward depend on what ran before it (stale registers / LDS)?  Same call, different preceding kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umnn_amd import IntegrandNetwork, integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(3)
for hid, B, d, E, n in (([50, 50, 50], 100, 3, 8, 20), ([50] * 4, 300, 6, 30, 100), ([40, 40], 64, 5, 4, 30)):
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g, gf = (torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev),
                   torch.randn(B, d, device=dev))
    ref = I.hip_backward(spec, None, x, h, g, gf, n)
    worst = 0.0
    for trial in range(6):
        junk = torch.randn(4096, 4096, device=dev) * (10.0 ** trial)
        (junk @ junk).sum().item()                       # dirty registers / LDS with other data
        if trial % 2:
            big = IntegrandNetwork(7, 31, [50] * 4, 1).to(dev)
            I.hip_forward(mlp_spec(big), None, torch.randn(2000, 7, device=dev) * 50, torch.randn(2000, 210, device=dev) * 50, 30)
        out = I.hip_backward(spec, None, x, h, g, gf, n)
        for a, b in zip(out, ref):
            worst = max(worst, float((a - b).abs().max()))
    print(hid, "max abs difference between identical calls in different contexts:", worst)
