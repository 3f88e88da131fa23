#!/usr/bin/env python
"""Energy per tile-node of the C3 forward launch under micro-variants of the kernel -- run on the GPU box:

    git apply tools/fwd_energy_probes.patch      (the probe switches are not kept in the product header)
    tools/build_fwd_variant.sh fnr1 -DUMNN_FWD_EXP_NORELOAD=1; ... fnr2 ...=2; ... fdup -DUMNN_FWD_EXP_DUPPAD      (build container)
    git checkout umnn_amd/csrc/cc_fwd_bf16_kernel.h
    python tools/fwd_energy_table.py [--seconds 5] [--out gpurun_out/fwd_energy.json]

One child process per library build (UMNN_CC_LIB), each launching the C3 forward (8192 x 63 integrals, n = 100, 31-50^4-1, f16x3)
back to back for `seconds` while tools/telemetry.Sampler polls the SMU: ms per launch, mean socket power, mean shader clock, and
from them joules per launch and nanojoules per tile-node (a tile-node = 16 integrals x one quadrature node).  Cases:
  as_is          the shipped kernel, default-initialised weights
  zero_weights   the shipped kernel, every hidden weight and bias zero (operand toggling of the weight side gone)
  big_weights    the shipped kernel, weights x 3 (more activation bits toggling; results stay finite)
  no_reload_l2   PROBE BUILD (results wrong): layer 2 runs on layer 1's fragments -- a third of the LDS fragment reads gone, what
                 holding one layer's weights in registers would save
  no_reload_all  PROBE BUILD (results wrong): no fragment is ever re-read after the prologue
  dup_padding    PROBE BUILD (results wrong): zero-padded rows / k-slots of the weight images hold copies of live weights
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, d, E, HID, n = 8192, 63, 30, [50] * 4, 100


def child(mode, seconds):
    import torch
    sys.path.insert(0, ROOT)
    from tools.telemetry import Sampler
    from umnn_amd import IntegrandNetwork, _lib
    from umnn_amd import integral as I
    from umnn_amd.nets import mlp_spec
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = IntegrandNetwork(d, 1 + E, HID, 1)
    with torch.no_grad():
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                if mode == "zero":
                    m.weight.zero_(); m.bias.zero_()
                if mode == "big":
                    m.weight.mul_(3.0)
    net.to(dev)
    spec = mlp_spec(net)
    x, h = torch.randn(B, d, device=dev), torch.randn(B, E * d, device=dev)
    sampler = Sampler(0)

    def fwd():
        I.hip_forward(spec, None, x, h, n)
    for _ in range(20):
        fwd()
    torch.cuda.synchronize()
    sampler.start()
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fwd()
        launches += 20
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rec = sampler.stop(skip_s=1.0) if sampler.ok else {}
    rec = {k: rec.get(k) for k in ("power_w", "power_w_max", "power_cap_w", "sclk_mhz", "sclk_mhz_min")}
    ms = 1e3 * wall / launches
    rec.update({"launches": launches, "ms_per_launch": ms, "kernel": _lib.lib().umnn_last_kernel_name().decode()})
    if rec.get("power_w"):
        tile_nodes = (B * d / 16) * (n + 1)
        rec["joules_per_launch"] = rec["power_w"] * ms * 1e-3
        rec["nJ_per_tile_node"] = 1e9 * rec["joules_per_launch"] / tile_nodes
        rec["Mcycles_per_launch"] = ms * 1e-3 * rec["sclk_mhz"]
    print("RESULT " + json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--out", default="gpurun_out/fwd_energy.json")
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.seconds)
    cases = [("as_is", "", "plain"), ("zero_weights", "", "zero"), ("big_weights", "", "big"),
             ("no_reload_l2", "fnr1", "plain"), ("no_reload_all", "fnr2", "plain"), ("dup_padding", "fdup", "plain"),
             ("as_is_again", "", "plain")]
    out = {"shape": {"rows": B, "dim": d, "E": E, "hidden": HID, "n_steps": n}, "cases": {}}
    for name, lib, mode in cases:
        env = dict(os.environ)
        if lib:
            path = os.path.join(ROOT, "umnn_amd", f"libumnn_cc_{lib}.so")
            if not os.path.exists(path):
                out["cases"][name] = {"skipped": f"{path} not built"}
                continue
            env["UMNN_CC_LIB"] = path
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, "--seconds", str(a.seconds)], env=env,
                           capture_output=True, text=True, timeout=300)
        line = next((l for l in r.stdout.splitlines() if l.startswith("RESULT ")), None)
        out["cases"][name] = json.loads(line[7:]) if line else {"error": (r.stderr or r.stdout)[-400:]}
        print(name, json.dumps(out["cases"][name]), flush=True)
        time.sleep(2.0)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
