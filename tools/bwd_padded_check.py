"""Backward of integrand nets with UNEQUAL hidden widths of 5..7 tiles (zero-padded onto the shape-exact fp32 families) against
the materialised ATen chain in float64 and float32; timing of both routes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import umnn_amd
from umnn_amd import _lib
from umnn_amd.nets import IntegrandNetwork
from umnn_amd.integral import ParallelNeuralIntegral, _flatten, mlp_spec, _hip_backward_ok

dev = torch.device("cuda:0")
for hid, E, B, d, n in [([100, 80, 70], 10, 64, 8, 20), ([70, 90], 30, 100, 16, 50), ([64, 100, 64, 100], 5, 33, 7, 30),
                        ([67, 65, 66], 12, 50, 10, 20), ([100, 50, 50, 50, 50], 30, 32, 49, 50), ([103, 97, 100], 30, 256, 49, 50), ([100, 100, 100], 30, 256, 49, 50), ([103, 97, 100], 30, 32, 49, 50), ([100, 100, 100], 30, 32, 49, 50), ([90, 100, 80], 30, 256, 49, 50), ([120, 112, 116], 10, 64, 8, 20), ([127, 70], 30, 100, 16, 50), ([110, 90, 127], 30, 256, 49, 50)]:
    torch.manual_seed(len(hid) + B)
    from umnn_amd.nets import IntegrandNN
    f = IntegrandNN(1 + E, hid).to(dev)
    x0 = torch.zeros(B * d, 1, device=dev)
    x = torch.randn(B * d, 1, device=dev) * 2
    h = torch.randn(B * d, E, device=dev)
    g = torch.randn(B * d, 1, device=dev)
    spec = mlp_spec(f)
    print(hid, "kind", _hip_backward_ok(spec, x, h), end="  ")

    def grads(module, dtype, generic):
        m = module.double() if dtype == torch.float64 else module
        xs = [t.detach().to(dtype).requires_grad_(True) for t in (x0, x, h)]
        from umnn_amd.integral import force_generic
        import contextlib
        with (force_generic() if generic else contextlib.nullcontext()):
            out = ParallelNeuralIntegral.apply(xs[0], xs[1], m, _flatten(m.parameters()), xs[2], n)
        gs = torch.autograd.grad(out, [xs[1], xs[2]] + list(m.parameters()), g.to(dtype))
        return torch.cat([t.reshape(-1).double() for t in gs])

    t0 = time.time(); hip = grads(f, torch.float32, False); torch.cuda.synchronize(); t_hip = time.time() - t0
    name = _lib.lib().umnn_last_kernel_name().decode()
    t0 = time.time(); hip = grads(f, torch.float32, False); torch.cuda.synchronize(); t_hip = time.time() - t0
    t0 = time.time(); gen32 = grads(f, torch.float32, True); torch.cuda.synchronize(); t_gen = time.time() - t0
    import copy
    truth = grads(copy.deepcopy(f), torch.float64, True)
    sc = truth.abs().max().item()
    print(f"hip {t_hip*1e3:7.2f} ms  aten {t_gen*1e3:7.2f} ms   |hip-truth| {((hip - truth).abs().max().item() / sc):.2e}   |aten32-truth| {((gen32 - truth).abs().max().item() / sc):.2e}   {name}")
