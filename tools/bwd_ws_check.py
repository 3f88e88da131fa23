#!/usr/bin/env python
"""Weight-stationary backward (UMNN_BWD_WS=1) against the software-pipelined loop and the exact-fp32 kernels, plus timing at C3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd  # noqa: E402
from umnn_amd import _lib  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
CASES = [(300, 63, 30, [50] * 4, 20, True), (2100, 8, 10, [50] * 4, 15, False), (301, 63, 30, [48, 60, 36, 50], 12, True),
         (1100, 16, 6, [50] * 4, 7, True)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    CASES = CASES[:1]
for (B, d, E, hid, n, gfx) in CASES:
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
    for key, ws, prec in (("swp", 0, "bf16x3"), ("ws", 1, "bf16x3"), ("ws16", 2, "bf16x3"), ("fp32", 0, "fp32")):
        _lib.set_backward_precision(prec)
        with _lib.options(bwd_ws=int(ws > 0), bwd_ws16=2 * int(ws == 2)):
            outs[key] = I.hip_backward(spec, x0, x, h, gg, gf, n)
            torch.cuda.synchronize()
        print(key, _lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode(), flush=True)
    _lib.set_backward_precision("bf16x3")
    for name, i in (("dx0", 0), ("dx", 1), ("dh", 2), ("dtheta", 3)):
        a, b, c, r = outs["swp"][i], outs["ws"][i], outs["ws16"][i], outs["fp32"][i]
        sc = float(r.abs().max())
        print(f"  {hid} B={B} {name:7s} |ws-swp|/max {float((a - b).abs().max()) / sc:.3e}   |swp-fp32| {float((a - r).abs().max()) / sc:.3e}"
              f"   |ws-fp32| {float((b - r).abs().max()) / sc:.3e}   |ws16-fp32| {float((c - r).abs().max()) / sc:.3e}   nan {bool(torch.isnan(c).any())}", flush=True)
    dd = (outs["fp32"][3] - outs["ws16"][3]).abs()
    o = 0
    for li, l in enumerate(spec.linears):
        a0, a1, a2 = o, o + l.weight.numel(), o + l.weight.numel() + l.bias.numel()
        o = a2
        print(f"     layer {li}: dW diff {float(dd[a0:a1].max()):.3e} db diff {float(dd[a1:a2].max()):.3e}   (|dW| {float(outs['swp'][3][a0:a1].abs().max()):.3e})")

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
# timing at C3
B, d, E, hid, n = 8192, 63, 30, [50] * 4, 100
torch.manual_seed(0)
net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
spec = mlp_spec(net)
x, h, g = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
gf = torch.randn(B, d, device=dev)
for ws in (1, 2, 1, 2):
    with _lib.options(bwd_ws=int(ws > 0), bwd_ws16=2 * int(ws == 2)):
        I.hip_backward(spec, None, x, h, g, gf, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            I.hip_backward(spec, None, x, h, g, gf, n)
        e1.record()
        torch.cuda.synchronize()
        print(f"C3 backward call (main + finishing) ws={ws}: {e0.elapsed_time(e1) / 5:.3f} ms   {_lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode()}", flush=True)
