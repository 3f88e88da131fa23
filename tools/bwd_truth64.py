#!/usr/bin/env python
"""Every backward kernel of the four-hidden-layer family against a FLOAT64 run of the reference's algorithm (materialised
nodes, torch autograd on the GPU; ParallelNeuralIntegral.py:66-94,110-123): which kernel is how far from the exact gradient at
batch sizes where the float32 routes differ from each other by LeakyReLU kink decisions."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd  # noqa: E402
from umnn_amd import _lib  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402
from umnn_amd.quadrature import compute_cc_weights  # noqa: E402

dev = torch.device("cuda:0")


def truth64(net, x0, x, h, g, gf, n):
    net64 = copy.deepcopy(net).double()
    w, s = compute_cc_weights(n)
    w, s = w.to(dev).double().reshape(-1), s.to(dev).double().reshape(-1)
    x0, x, g = x0.double(), x.double(), g.double()
    h = h.double().requires_grad_()
    B, d = x.shape
    t = x0[:, None, :] + (x - x0)[:, None, :] * (s[None, :, None] + 1) / 2            # [B, n+1, d]
    hs = h[:, None, :].expand(B, n + 1, h.shape[1]).reshape(B * (n + 1), -1)
    f = net64(t.reshape(B * (n + 1), d), hs).reshape(B, n + 1, d)
    cot = (g * (x - x0) / 2)[:, None, :] * w[None, :, None]
    loss = (cot * f).sum()
    xr = x.clone().requires_grad_()
    fx = net64(xr, h)
    if gf is not None:
        loss = loss + (gf.double() * fx).sum()
    params = list(net64.parameters())
    grads = torch.autograd.grad(loss, params + [h] + ([xr] if gf is not None else []), allow_unused=True)
    dtheta = torch.cat([gr.reshape(-1) for gr in grads[:len(params)]])
    dh = grads[len(params)]
    dx = fx.detach() * g + (grads[-1] if gf is not None else 0)
    dx0 = -net64(x0, h).detach() * g
    return dx0, dx, dh, dtheta


CASES = [(300, 63, 30, [50] * 4, 20, True, 1.7), (2100, 8, 10, [50] * 4, 15, False, 1.7), (1100, 16, 6, [50] * 4, 7, True, 1.7),
         (1024, 63, 30, [50] * 4, 20, True, 1.0), (2048, 6, 30, [50] * 4, 100, True, 1.0)]
for (B, d, E, hid, n, gfx, scale) in CASES:
    torch.manual_seed(B * 7 + d)
    net = umnn_amd.IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.mul_(scale)
    spec = mlp_spec(net)
    x, x0 = torch.randn(B, d, device=dev) * 2, torch.randn(B, d, device=dev) * 0.3
    h, gg = torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev) if gfx else None
    ref = truth64(net, x0, x, h, gg, gf, n)
    line = f"B={B} d={d} n={n} x{scale}:"
    for key, ws, ws16, prec in (("swp", 0, 0, "bf16x3"), ("ws", 1, 0, "bf16x3"), ("ws16", 1, 1, "bf16x3"), ("fp32", 0, 0, "fp32")):
        _lib.set_backward_precision(prec)
        with _lib.options(bwd_ws=ws, bwd_ws16=2 * ws16):
            out = I.hip_backward(spec, x0, x, h, gg, gf, n)
            torch.cuda.synchronize()
        _lib.set_backward_precision("bf16x3")
        errs = [float((o.double() - r).abs().max() / r.abs().max()) for o, r in zip(out, ref)]
        line += f"\n    {key:5s} {_lib.lib().umnn_last_kernel_name_of(_lib.PROF_BACKWARD).decode():34s} dx0 {errs[0]:.1e} dx {errs[1]:.1e} dh {errs[2]:.1e} dtheta {errs[3]:.1e}"
    print(line, flush=True)
