#!/usr/bin/env python
"""Socket power and shader clock under the flagship kernels (run on the GPU box):

  python tools/power_probe.py [--seconds 6] [--out gpurun_out/power_fwd.json]

For each case -- idle, the C3 forward launch with fwd_pipe = 1 (16x16x32 pipelined loop, the default) and fwd_pipe = 2 (the
32x32x16 formulation), the exact-fp32 forward, the C3 backward -- the kernel is launched back to back for >= ``seconds``
while tools/telemetry.Sampler polls the SMU metrics table; the first second (ramp) is dropped.  The JSON holds, per case:
launches, ms per launch (wall / launches, launches are back to back), mean / max socket power, the power cap, mean / min
shader clock.  This is the direct evidence DESIGN 9.3's "the forward kernel is power-bound" claim needs (or refutes).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.telemetry import Sampler  # noqa: E402
from umnn_amd import IntegrandNetwork, _lib  # noqa: E402
from umnn_amd import integral as I  # noqa: E402
from umnn_amd.nets import mlp_spec  # noqa: E402

B, d, E, HID, n = 8192, 63, 30, [50] * 4, 100


def loop(fn, seconds, sampler, skip=1.0):
    fn()
    torch.cuda.synchronize()
    sampler.start()
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        launches += 20
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rec = sampler.stop(skip_s=skip)
    rec.update({"launches": launches, "ms_per_launch": 1e3 * wall / launches})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--out", default="gpurun_out/power_fwd.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = IntegrandNetwork(d, 1 + E, HID, 1).to(dev)
    spec = mlp_spec(net)
    x, h, g = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev), torch.randn(B, d, device=dev)
    sampler = Sampler(0)
    out = {"device": torch.cuda.get_device_name(0), "shape": {"rows": B, "dim": d, "E": E, "hidden": HID, "n_steps": n},
           "sampler": {"source": sampler.src.name if sampler.ok else None, "errors": sampler.errors}, "cases": {}}
    if not sampler.ok:
        print(json.dumps(out))
        return

    sampler.start()
    time.sleep(2.0)
    out["cases"]["idle"] = sampler.stop()
    lib = _lib.lib()

    def fwd():
        I.hip_forward(spec, None, x, h, n)

    def bwd():
        I.hip_backward(spec, None, x, h, g, None, n)

    for name, opts, fn in (("fwd_bf16x3_pipe1", {"fwd_precision": "bf16x3", "fwd_pipe": 1}, fwd),
                           ("fwd_bf16x3_pipe2_32x32x16", {"fwd_precision": "bf16x3", "fwd_pipe": 2}, fwd),
                           ("fwd_bf16x3_plain_loop", {"fwd_precision": "bf16x3", "fwd_pipe": 0}, fwd),
                           ("fwd_fp32", {"fwd_precision": "fp32", "fwd_pipe": 1}, fwd),
                           ("bwd_bf16x3", {"fwd_precision": "bf16x3", "fwd_pipe": 1}, bwd)):
        _lib.set_forward_precision(opts["fwd_precision"])
        _lib.set_option("fwd_pipe", opts["fwd_pipe"])
        rec = loop(fn, a.seconds, sampler)
        rec["kernel"] = lib.umnn_last_kernel_name().decode()
        out["cases"][name] = rec
        print(name, json.dumps(rec), flush=True)
        time.sleep(1.0)
    _lib.set_forward_precision("bf16x3")
    _lib.set_option("fwd_pipe", 1)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
