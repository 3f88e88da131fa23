#!/usr/bin/env python
"""Kernel-only timing of umnn_cc_forward (hipEvents on the launch stream, via umnn_cc_forward_timed).

  python tools/fwd_sweep.py [--shape bsds300|power|toy|vae] [--reps 10]
Env overrides UMNN_FWD_P / UMNN_FWD_NS select launch heuristics.  Prints ms, TFLOP/s (algorithmic), % of fp32 MFMA peak.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umnn_amd import _lib, IntegrandNetwork  # noqa: E402
from umnn_amd.integral import _desc, _ptr  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402
from umnn_amd.quadrature import device_tables  # noqa: E402

SHAPES = {"bsds300": (8192, 63, 30, [50] * 4, 100), "power": (10000, 6, 30, [50] * 4, 100),
          "toy": (4096, 2, 10, [100] * 4, 50), "vae": (1024, 64, 30, [50] * 4, 50),
          "mnist": (256, 784, 30, [100, 50, 50, 50, 50], 50)}


def run(shape, reps, override=None):
    B, d, E, hid, n = override or SHAPES[shape]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = IntegrandNetwork(d, 1 + E, hid, 1).to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
    F, fx, fx0 = (torch.empty_like(x) for _ in range(3))
    w, s = device_tables(n, dev)
    desc, keep = _desc(spec)
    lib = _lib.lib()
    ms = ctypes.c_float()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.umnn_cc_forward_timed(ctypes.byref(desc), None, _ptr(x), _ptr(h), _ptr(w), _ptr(s), n, B, d, E,
                                   _ptr(F), _ptr(fx), _ptr(fx0), reps, ctypes.byref(ms), stream)
    _lib.check(rc, "forward_timed")
    fl = lib.umnn_cc_forward_flops_per_integral(ctypes.byref(desc), n) * B * d
    tf = fl / (ms.value * 1e-3) / 1e12
    print(f"{shape:8s} P={os.environ.get('UMNN_FWD_P','auto'):4s} NS={os.environ.get('UMNN_FWD_NS','auto'):4s} "
          f"{lib.umnn_last_kernel_name().decode():28s} {ms.value:8.3f} ms  {tf:7.2f} TFLOP/s  {100*tf/157.3:5.1f}% of fp32 MFMA peak",
          flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="bsds300")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--custom", default="", help="B,d,E,H,L,n  e.g. 1536,64,30,50,4,100")
    a = ap.parse_args()
    if a.custom:
        B, d, E, H, L, n = (int(v) for v in a.custom.split(","))
        run(f"B{B}d{d}", a.reps, (B, d, E, [H] * L, n))
    else:
        for sh in a.shape.split(","):
            run(sh, a.reps)
