#!/usr/bin/env python
"""Per-output difference of the two one-pass backward node loops (UMNN_BWD_SWP 0 / 1) and of each against the exact-fp32 kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd  # noqa: E402
from umnn_amd import _lib  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
for (B, d, E, hid, n, gfx) in [(257, 63, 30, [50] * 4, 100, False), (64, 5, 8, [50] * 2, 20, True), (5, 3, 4, [48, 60, 36], 7, True),
                               (100, 6, 30, [50] * 3, 50, False)]:
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(1.7)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev) if gfx else None
    outs = {}
    for key, swp, prec in (("old", 0, "bf16x3"), ("swp", 1, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        with _lib.options(bwd_swp=swp):
            outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
            torch.cuda.synchronize()
        print(key, _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode())
    _lib.set_backward_precision("bf16x3")
    for name, i in (("dx0", 0), ("dx", 1), ("dh", 2), ("dtheta", 3)):
        a, b, r = outs["old"][i], outs["swp"][i], outs["fp32"][i]
        sc = float(r.abs().max())
        print(f"  {hid} B={B} {name:7s} |swp-old|/max {float((a - b).abs().max()) / sc:.3e}   |old-fp32| {float((a - r).abs().max()) / sc:.3e}"
              f"   |swp-fp32| {float((b - r).abs().max()) / sc:.3e}   nan {bool(torch.isnan(b).any())}")
    if not torch.equal(outs["old"][3], outs["swp"][3]):
        dd = (outs["old"][3] - outs["swp"][3]).abs()
        offs, o = [], 0
        for l in spec.linears:
            offs.append((o, o + l.weight.numel(), o + l.weight.numel() + l.bias.numel()))
            o += l.weight.numel() + l.bias.numel()
        for li, (a0, a1, a2) in enumerate(offs):
            print(f"     layer {li}: dW diff {float(dd[a0:a1].max()):.3e} db diff {float(dd[a1:a2].max()):.3e}   (|dW| {float(outs['old'][3][a0:a1].abs().max()):.3e})")
