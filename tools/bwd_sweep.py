#!/usr/bin/env python
"""Timing of the HIP backward (umnn_cc_backward: main pass(es) + finishing kernels) with events on the launch stream."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umnn_amd import IntegrandNetwork, _lib  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402
from tools.fwd_sweep import SHAPES  # noqa: E402


def run(shape, reps, gfx, override=None):
    B, d, E, hid, n = override or SHAPES[shape]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    gf = torch.randn(B, d, device=dev) if gfx else None
    I.hip_backward(spec, None, x, h, g, gf, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        I.hip_backward(spec, None, x, h, g, gf, n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    desc, keep = I._desc(spec)
    import ctypes
    fl = _lib.lib().umnn_cc_forward_flops_per_integral(ctypes.byref(desc), n) * B * d
    print(f"{shape:8s} gfx={int(gfx)} backward {ms:8.3f} ms  ({3.2 * fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s at 3.2x forward FLOPs)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="bsds300")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--custom", default="", help="B,d,E,H,L,n  e.g. 100,2,10,100,4,20")
    a = ap.parse_args()
    if a.custom:
        B, d, E, H, L, n = (int(v) for v in a.custom.split(","))
        run(f"B{B}d{d}", a.reps, True, (B, d, E, [H] * L, n))
    else:
        for sh in a.shape.split(","):
            run(sh, a.reps, False)
            run(sh, a.reps, True)
