"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  PYTHONPATH=/root/reference python tests/golden/make_golden.py
Writes small .npz fixtures next to this file.  The reference never travels to
the GPU box; these fixtures (data only: seeded inputs, the reference's own
default-initialised weights, and the outputs the reference computed from them
on this container's torch CPU build) are what pins the oracle and the HIP path.

Every array is produced by calling reference code:
  models/UMNN/ParallelNeuralIntegral.py  compute_cc_weights, integrate, ParallelNeuralIntegral
  models/UMNN/NeuralIntegral.py          integrate, NeuralIntegral
  models/UMNN/UMNNMAF.py                 IntegrandNetwork, EmbeddingNetwork, UMNNMAF
  models/UMNN/UMNNMAFFlow.py             UMNNMAFFlow
  models/UMNN/MonotonicNN.py             MonotonicNN
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from models.UMNN import UMNNMAFFlow, MonotonicNN  # noqa: E402
from models.UMNN.UMNNMAF import IntegrandNetwork  # noqa: E402
import importlib  # noqa: E402
NI_mod = importlib.import_module("models.UMNN.NeuralIntegral")  # the module, not the re-exported class
PNI_mod = importlib.import_module("models.UMNN.ParallelNeuralIntegral")
from models.UMNN.NeuralIntegral import NeuralIntegral  # noqa: E402
from models.UMNN.ParallelNeuralIntegral import ParallelNeuralIntegral  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def flat(ps):
    return torch.cat([p.contiguous().view(-1) for p in ps])


def save(name, **arrs):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, {k: v.shape for k, v in list(arrs.items())[:6]}, "...")


# ---------------------------------------------------------------- G1 tables
def g1():
    arrs = {}
    for n in (5, 7, 20, 30, 50, 51, 100, 200):
        w, s = PNI_mod.compute_cc_weights(n)
        arrs[f"w{n}"], arrs[f"s{n}"] = w, s
    save("g1_cc_tables", **arrs)


# ------------------------------------------------------- G2/G3 integrate + grads
CASES = [
    # name, d, E, hidden, n, B, act, weight scale, x0 nonzero
    ("toy_d2", 2, 10, [100] * 4, 50, 16, "ELU", 1.0, False),
    ("toy_d2_w2", 2, 10, [100] * 4, 50, 16, "ELU", 2.0, True),
    ("power_d6", 6, 30, [50] * 4, 100, 16, "ELU", 1.0, False),
    ("power_d6_w2", 6, 30, [50] * 4, 100, 16, "ELU", 2.0, True),
    ("bsds_d63", 63, 30, [50] * 4, 100, 4, "ELU", 1.0, False),
    ("bsds_d63_w2", 63, 30, [50] * 4, 100, 3, "ELU", 2.0, True),
    ("vae_d64", 64, 30, [50] * 4, 50, 4, "ELU", 1.5, False),
    ("jit_d5", 5, 1, [50, 50], 20, 10, "ELU", 1.0, False),
    ("fd_d3", 3, 1, [20, 20], 20, 10, "ELU", 3.0, True),
    ("sigmoid_d4", 4, 3, [30, 30, 30], 25, 8, "Sigmoid", 2.0, True),
    ("mnist_mixed_d8", 8, 30, [100, 50, 50, 50, 50], 50, 4, "ELU", 1.5, False),
    ("odd_n_d3", 3, 2, [16], 7, 5, "ELU", 3.0, True),
]


def g23():
    for seed, (name, d, E, hid, n, B, act, wscale, x0nz) in enumerate(CASES):
        torch.manual_seed(1000 + seed)
        net = IntegrandNetwork(d, 1 + E, hid, 1, act_func=act)
        with torch.no_grad():
            for p in net.net:
                if isinstance(p, torch.nn.Linear):
                    p.weight.mul_(wscale)
                    p.bias.mul_(wscale)
        x = torch.randn(B, d) * 2.0
        x0 = torch.randn(B, d) * 0.7 if x0nz else torch.zeros(B, d)
        h = torch.randn(B, E * d)
        g = torch.randn(B, d)
        arrs = dict(d=d, E=E, n=n, hidden=np.array(hid), act=act, x=x, x0=x0, h=h, g=g)
        lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
        for l, m in enumerate(lin):
            arrs[f"W{l}"], arrs[f"b{l}"] = m.weight, m.bias
        with torch.no_grad():
            arrs["F_par"] = PNI_mod.integrate(x0, n, (x - x0) / n, net, h, False)
            arrs["F_seq"] = NI_mod.integrate(x0, n, (x - x0) / n, net, h, False)
            arrs["F_inv"] = PNI_mod.integrate(x0, n, (x - x0) / n, net, h, False, None, True)
            arrs["f_x"] = net(x, h)
            arrs["f_x0"] = net(x0, h)
        for tag, Fn, extra in (("par", ParallelNeuralIntegral, (False,)), ("seq", NeuralIntegral, ())):
            net.zero_grad()
            x0r, xr, hr = x0.clone().requires_grad_(), x.clone().requires_grad_(), h.clone().requires_grad_()
            out = Fn.apply(x0r, xr, net, flat(net.parameters()), hr, n, *extra)
            out.backward(g)
            arrs[f"dx0_{tag}"], arrs[f"dx_{tag}"], arrs[f"dh_{tag}"] = x0r.grad, xr.grad, hr.grad
            arrs[f"dtheta_{tag}"] = flat([p.grad for p in net.parameters()])
            arrs[f"Fapply_{tag}"] = out
        print("   F range", float(arrs["F_par"].min()), float(arrs["F_par"].max()),
              "f range", float(arrs["f_x"].min()), float(arrs["f_x"].max()))
        save("g2_" + name, **arrs)


# ------------------------------------------------------------------ G4 flows
def sd_arrays(model):
    return {"sd/" + k: v for k, v in model.state_dict().items()}


def g4():
    cfgs = [
        # name, nb_flow, d, hid_int, hid_emb, E, n, solver, cond_in, B
        ("flow1_power", 1, 6, [50] * 4, [64, 64], 30, 100, "CCParallel", 0, 12),
        ("flow2_power_cc", 2, 6, [50] * 4, [64, 64], 30, 50, "CC", 0, 12),
        ("flow2_toy", 2, 2, [100] * 4, [40, 40], 10, 50, "CCParallel", 0, 32),
        ("flow2_cond", 2, 4, [50] * 4, [48, 48], 30, 50, "CCParallel", 3, 10),
        ("flow3_d17", 3, 17, [50] * 4, [48, 48], 30, 30, "CCParallel", 0, 6),
    ]
    for seed, (name, nf, d, hi, he, E, n, solver, cond, B) in enumerate(cfgs):
        torch.manual_seed(2000 + seed)
        model = UMNNMAFFlow(nb_flow=nf, nb_in=d, hidden_derivative=hi, hidden_embedding=he,
                            embedding_s=E, nb_steps=n, solver=solver, cond_in=cond)
        # make the blocks non-trivial (default init gives f ~ 1 everywhere)
        with torch.no_grad():
            for i in range(nf):
                for m in model.nets[i].net.parallel_nets.net:
                    if isinstance(m, torch.nn.Linear):
                        m.weight.mul_(1.5)
        x = torch.randn(B, d)
        ctx = torch.randn(B, cond) if cond > 0 else None
        arrs = dict(nb_flow=nf, d=d, E=E, n=n, solver=solver, cond_in=cond, hidden_derivative=np.array(hi),
                    hidden_embedding=np.array(he), x=x)
        if ctx is not None:
            arrs["context"] = ctx
        arrs.update(sd_arrays(model))
        for mode in ("train", "eval"):
            model.train(mode == "train")
            with torch.no_grad():
                ll, z = model.compute_ll(x, context=ctx)
                arrs[f"ll_{mode}"], arrs[f"z_{mode}"] = ll, z
                arrs[f"log_jac_{mode}"] = model.compute_log_jac(x, context=ctx)
                zb, ljb = model.compute_log_jac_bis(x, context=ctx)
                arrs[f"z_bis_{mode}"], arrs[f"log_jac_bis_{mode}"] = zb, ljb
                arrs[f"fwd_{mode}"] = model.forward(x, context=ctx)
                llb, zb2 = model.compute_ll_bis(x, context=ctx)
                arrs[f"ll_bis_{mode}"] = llb
                bpp, _, _ = model.compute_bpp(x, context=ctx)
                arrs[f"bpp_{mode}"] = bpp
        # training gradients of -mean(ll) wrt every trainable parameter and x
        model.train()
        model.zero_grad()
        xr = x.clone().requires_grad_()
        ll, _ = model.compute_ll(xr, context=ctx)
        (-ll.mean()).backward()
        arrs["grad/x"] = xr.grad
        for k, p in model.named_parameters():
            if p.grad is not None:
                arrs["grad/" + k] = p.grad
        save("g4_" + name, **arrs)


# -------------------------------------------------------------- G5 MonotonicNN
def g5():
    for n in (50, 100):
        torch.manual_seed(3000 + n)
        model = MonotonicNN(3, [100, 100, 100], nb_steps=n, dev="cpu")
        with torch.no_grad():
            for m in model.integrand.net:
                if isinstance(m, torch.nn.Linear):
                    m.weight.mul_(1.5)
        x = torch.randn(100, 1)
        h = torch.randn(100, 2)
        arrs = dict(n=n, x=x, h=h)
        arrs.update(sd_arrays(model))
        xr = x.clone().requires_grad_()
        y = model(xr, h)
        arrs["y"] = y
        (y ** 2).mean().backward()
        arrs["grad/x"] = xr.grad
        for k, p in model.named_parameters():
            arrs["grad/" + k] = p.grad
        save(f"g5_monotonic_n{n}", **arrs)


# ------------------------------------------------------------------ G6 invert
def g6():
    torch.manual_seed(4000)
    model = UMNNMAFFlow(nb_flow=2, nb_in=2, hidden_derivative=[50] * 3, hidden_embedding=[32, 32],
                        embedding_s=10, nb_steps=30, solver="CCParallel")
    model.eval()
    x = torch.randn(8, 2)
    with torch.no_grad():
        z = model.forward(x)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            x_inv = model.invert(z, iter=5)
    arrs = dict(x=x, z=z, x_inv=x_inv)
    arrs.update(sd_arrays(model))
    print("   invert max err", float((x_inv - x).abs().max()))
    save("g6_invert", **arrs)


# ------------------------------------------------- G7 inv_f = True through the operator (forward + custom backward)
INV_CASES = [
    # name, d, E, hidden, n, B, act, weight scale, x0 nonzero
    ("power_d6", 6, 30, [50] * 4, 100, 12, "ELU", 2.0, True),
    ("toy_d2", 2, 10, [100] * 4, 50, 12, "ELU", 1.5, False),
    ("mnist_mixed_d8", 8, 30, [100, 50, 50, 50, 50], 50, 4, "ELU", 1.5, False),
    ("sigmoid_d4", 4, 3, [30, 30, 30], 25, 8, "Sigmoid", 2.0, True),
    ("narrow_d3", 3, 2, [20, 20], 20, 10, "ELU", 1.5, True),
]


def g7():
    """ParallelNeuralIntegral.apply(..., inv_f=True) and its backward (ParallelNeuralIntegral.py:58-59,70-72,110-123): the
    parameter / embedding gradients differentiate 1/f, the Leibniz terms keep f (both branches of :120-123 are the same)."""
    for seed, (name, d, E, hid, n, B, act, wscale, x0nz) in enumerate(INV_CASES):
        torch.manual_seed(7000 + seed)
        net = IntegrandNetwork(d, 1 + E, hid, 1, act_func=act)
        with torch.no_grad():
            for p in net.net:
                if isinstance(p, torch.nn.Linear):
                    p.weight.mul_(wscale)
                    p.bias.mul_(wscale)
        x = torch.randn(B, d) * 2.0
        x0 = torch.randn(B, d) * 0.7 if x0nz else torch.zeros(B, d)
        h = torch.randn(B, E * d)
        g = torch.randn(B, d)
        arrs = dict(d=d, E=E, n=n, hidden=np.array(hid), act=act, x=x, x0=x0, h=h, g=g)
        lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
        for l, m in enumerate(lin):
            arrs[f"W{l}"], arrs[f"b{l}"] = m.weight, m.bias
        net.zero_grad()
        x0r, xr, hr = x0.clone().requires_grad_(), x.clone().requires_grad_(), h.clone().requires_grad_()
        out = ParallelNeuralIntegral.apply(x0r, xr, net, flat(net.parameters()), hr, n, True)
        out.backward(g)
        arrs["F_inv"], arrs["dx0"], arrs["dx"], arrs["dh"] = out, x0r.grad, xr.grad, hr.grad
        arrs["dtheta"] = flat([p.grad for p in net.parameters()])
        save("g7_invf_" + name, **arrs)


def g8():
    """One case at a size the weight-stationary workgroup-pipeline backward takes (>= 4 tiles of 16 integrals per workgroup on 256
    CUs: 280 x 63 = 17 640 integrals), so that a REFERENCE-produced d_theta / d_h exists at that size:
    ParallelNeuralIntegral.apply(...).backward(g), ParallelNeuralIntegral.py:97-123."""
    name, d, E, hid, n, B, act, wscale = "ws_d63", 63, 10, [50] * 4, 20, 280, "ELU", 1.5
    torch.manual_seed(8000)
    net = IntegrandNetwork(d, 1 + E, hid, 1, act_func=act)
    with torch.no_grad():
        for p in net.net:
            if isinstance(p, torch.nn.Linear):
                p.weight.mul_(wscale)
                p.bias.mul_(wscale)
    x, x0 = torch.randn(B, d) * 2.0, torch.randn(B, d) * 0.7
    h, g = torch.randn(B, E * d), torch.randn(B, d)
    arrs = dict(d=d, E=E, n=n, hidden=np.array(hid), act=act, x=x, x0=x0, h=h, g=g)
    lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
    for l, m in enumerate(lin):
        arrs[f"W{l}"], arrs[f"b{l}"] = m.weight, m.bias
    net.zero_grad()
    x0r, xr, hr = x0.clone().requires_grad_(), x.clone().requires_grad_(), h.clone().requires_grad_()
    out = ParallelNeuralIntegral.apply(x0r, xr, net, flat(net.parameters()), hr, n, False)
    out.backward(g)
    arrs["F_par"], arrs["dx0_par"], arrs["dx_par"], arrs["dh_par"] = out, x0r.grad, xr.grad, hr.grad
    arrs["dtheta_par"] = flat([p.grad for p in net.parameters()])
    save("g8_" + name, **arrs)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["g1", "g23", "g4", "g5", "g6", "g7", "g8"]      # (name a subset to leave the other fixtures untouched)
    for name in todo:
        globals()[name]()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("total fixture bytes", tot)
