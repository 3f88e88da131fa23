"""Test-side evaluator of the reference's custom backward in an arbitrary torch dtype, chunked over rows, on any device.

TEST INFRASTRUCTURE (like oracle/): only tests/ and tools/ import it; nothing under umnn_amd/ does.  It restates
``ParallelNeuralIntegral.backward`` / ``integrate(compute_grad=True)`` (models/UMNN/ParallelNeuralIntegral.py:66-94,110-123):

    d_theta, d_h = VJP of the integrand over ALL quadrature nodes t_k = x0 + (x - x0)(s_k + 1)/2 with cotangent g (x - x0)/2 w_k
    d_x = f(x; h) g,   d_x0 = -f(x0; h) g                                            (Leibniz terms, :117-118)
    optional g_fx: + the VJP of f(x; h) itself with cotangent g_fx (d_x, d_h, d_theta) -- umnn_amd's IntegralWithJacobian output

with the node axis materialised exactly as the reference does (rows x (n+1) points through the integrand, torch autograd), which is
why it is chunked: C3 (8192 x 63 x 101 nodes) is 52 M points.  In ``torch.float64`` on the GPU box it is the TRUTH the -m gpu tests
hold the HIP backward to at the benchmarked sizes (numpy float64 on the host would take minutes); in ``torch.float32`` it is "the
reference's own arithmetic" (ATen fp32 GEMMs), i.e. how far a faithful float32 run of the reference itself sits from that truth.
Pinned by tests/test_gpu_round5.py::test_truth64_evaluator_matches_the_pinned_oracle (oracle.integrate_backward on the g8 reference
fixture, which the reference itself produced).
"""
import copy

import torch


def backward_reference(net, x0, x, h, g, g_fx, nb_steps, dtype=torch.float64, chunk=256, inv_f=False):
    """net: an nn.Module ``net(x_rows [R, d], h_rows [R, E*d]) -> [R, d]`` (umnn_amd.IntegrandNetwork or the reference's).
    -> (d_x0, d_x, d_h, d_theta_flat) in ``dtype`` on x's device; d_theta in the order of ``net.parameters()``."""
    from umnn_amd.quadrature import compute_cc_weights
    dev = x.device
    netd = copy.deepcopy(net).to(dev).to(dtype)
    params = [p for p in netd.parameters()]
    w, s = compute_cc_weights(nb_steps)
    w, s = w.to(dev).to(dtype).reshape(-1), s.to(dev).to(dtype).reshape(-1)
    B, d = x.shape
    n1 = nb_steps + 1
    x0 = torch.zeros_like(x) if x0 is None else x0
    dth = [torch.zeros_like(p) for p in params]
    dx0s, dxs, dhs = [], [], []
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        xa, x0a, ga = x[sl].to(dtype), x0[sl].to(dtype), g[sl].to(dtype)
        ha = h[sl].to(dtype).detach().requires_grad_()
        b = xa.shape[0]
        t = x0a[:, None, :] + (xa - x0a)[:, None, :] * (s[None, :, None] + 1) / 2            # [b, n+1, d]
        hs = ha[:, None, :].expand(b, n1, ha.shape[1]).reshape(b * n1, -1)
        f = netd(t.reshape(b * n1, d), hs).reshape(b, n1, d)
        if inv_f:
            f = 1.0 / f
        cot = (ga * (xa - x0a) / 2)[:, None, :] * w[None, :, None]
        loss = (cot * f).sum()
        xr = xa.clone().requires_grad_()
        fx = netd(xr, ha)
        wrt = params + [ha]
        if g_fx is not None:
            loss = loss + (g_fx[sl].to(dtype) * fx).sum()
            wrt = wrt + [xr]
        grads = torch.autograd.grad(loss, wrt, allow_unused=True)
        for acc, gr in zip(dth, grads[:len(params)]):
            if gr is not None:
                acc += gr
        dhs.append(grads[len(params)])
        fxd, fx0d = fx.detach(), netd(x0a, ha).detach()
        if inv_f:
            fxd, fx0d = 1.0 / fxd, 1.0 / fx0d
        dxs.append(fxd * ga + (grads[-1] if g_fx is not None else 0))
        dx0s.append(-fx0d * ga)
    return torch.cat(dx0s), torch.cat(dxs), torch.cat(dhs), torch.cat([a.reshape(-1) for a in dth])


def scaled_err(a, ref):
    """max |a - ref| / max |ref| (the gradient criterion of SURVEY 8d), in float64."""
    a, ref = a.double(), ref.double()
    return float((a - ref).abs().max() / ref.abs().max().clamp(min=1e-300))
