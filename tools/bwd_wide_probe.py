#!/usr/bin/env python
"""Deep wide integrands (four or more hidden layers above 103 units): the generic HIP backward kernels (what UMNN_BWD_WIDE=hip forces) against
the materialised ATen chain the library differentiates them with by default.    python tools/bwd_wide_probe.py    (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umnn_amd import IntegrandNetwork  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

dev = torch.device("cuda:0")
for (B, d, E, hid, n) in ((256, 6, 10, [110] * 4, 50), (2048, 6, 10, [110] * 4, 50), (256, 6, 10, [120, 110, 104, 112], 50), (512, 8, 10, [70] * 5, 50)):
    torch.manual_seed(0)
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g, gf = (torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev), torch.randn(B, d, device=dev))
    print(hid, "B", B, "hip_backward_ok:", I._hip_backward_ok(spec, x, h), flush=True)
    res = {}
    for name, fn in (("hip (generic kernels)", lambda: I.hip_backward(spec, None, x, h, g, gf, n)),
                     ("aten chain", lambda: I.aten_backward_jac(net, torch.zeros_like(x), x, h, g, gf, n))):
        try:
            out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            res[name] = ((time.perf_counter() - t0) / 3 * 1e3, out)
            print(f"   {name:24s} {res[name][0]:9.2f} ms", flush=True)
        except Exception as e:
            print(f"   {name:24s} failed: {type(e).__name__}: {str(e)[:120]}", flush=True)
    if len(res) == 2:
        a, b = res["hip (generic kernels)"][1], res["aten chain"][1]
        for nm, u, v in zip(("dx0", "dx", "dh", "dtheta"), a, b):
            if u is not None and v is not None:
                print(f"      {nm}: max |hip - aten| / max |aten| = {float((u - v).abs().max() / v.abs().max()):.2e}")
