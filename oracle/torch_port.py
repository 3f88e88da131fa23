"""PyTorch-CPU port of the reference's *materialise-all-nodes* algorithm -- the CPU baseline that is timed.

TEST / BENCH INFRASTRUCTURE ONLY (same rule as cc_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg
import this).  The reference itself is Python and cannot travel to the GPU box, so ``bench.py`` times this port on
the node's host cores as ``cpu_baseline.kind = "port"``.  It issues the same ATen operator sequence as the
reference so its cost profile (expand + contiguous copies of h over the node axis, transposed copy in the
integrand, (n+1)x activations) is the reference's:
  parallel quadrature   models/UMNN/ParallelNeuralIntegral.py:49-65
  sequential quadrature models/UMNN/NeuralIntegral.py:53-66
  integrand rows        models/UMNN/UMNNMAF.py:263-284
  block / flow          models/UMNN/UMNNMAF.py:76-139, models/UMNN/UMNNMAFFlow.py:109-119
Parity with the reference's golden vectors is checked in tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn.functional as F

from oracle.cc_oracle import cc_tables


def integrand(Ws, bs, x, h, d, sigmoid=False):
    B = x.shape[0]
    rows = torch.cat((x, h), 1).view(B, -1, d).transpose(1, 2).contiguous().view(B * d, -1)
    a = rows
    for l, (W, b) in enumerate(zip(Ws, bs)):
        a = F.linear(a, W, b)
        if l < len(Ws) - 1:
            a = F.leaky_relu(a, 0.01)
    a = torch.sigmoid(a) if sigmoid else F.elu(a) + 1.
    return a.view(B, -1)


def integrate_parallel(Ws, bs, x0, x, h, n, sigmoid=False):
    w, s = cc_tables(n)
    w, s = torch.from_numpy(w).view(-1, 1), torch.from_numpy(s).view(-1, 1)
    B, d = x.shape
    xT = x0 + n * ((x - x0) / n)
    x0_t = x0.unsqueeze(1).expand(-1, n + 1, -1)
    xT_t = xT.unsqueeze(1).expand(-1, n + 1, -1)
    h_steps = h.unsqueeze(1).expand(-1, n + 1, -1).contiguous().view(-1, h.shape[1])
    steps_t = s.unsqueeze(0).expand(B, -1, d)
    X = (x0_t + (xT_t - x0_t) * (steps_t + 1) / 2).contiguous().view(-1, d)
    f = integrand(Ws, bs, X, h_steps, d, sigmoid).view(B, n + 1, -1)
    return (f * w.unsqueeze(0)).sum(1) * (xT - x0) / 2


def integrate_sequential(Ws, bs, x0, x, h, n, sigmoid=False):
    w, s = cc_tables(n)
    d = x.shape[1]
    xT = x0 + n * ((x - x0) / n)
    z = 0.
    for k in range(n + 1):
        t = x0 + (xT - x0) * (float(s[k]) + 1) / 2
        z = z + float(w[k]) * integrand(Ws, bs, t, h, d, sigmoid)
    return z * (xT - x0) / 2


def made(Ws, bs, masks, x):
    a = x
    for l, (W, b, m) in enumerate(zip(Ws, bs, masks)):
        a = F.linear(a, m * W, b)
        if l < len(Ws) - 1:
            a = torch.relu(a)
    return a


def _embedding(blk, x, context, cond_in):
    """MADE pass of one block; with a context, the reference's ConditionnalMADE (made.py:165-168): MADE over
    [context, x], then every output chunk drops its cond_in context columns."""
    if context is None:
        return made(blk["mW"], blk["mb"], blk["mm"], x)
    out = made(blk["mW"], blk["mb"], blk["mm"], torch.cat((context, x), 1))
    B, nin = x.shape[0], x.shape[1] + cond_in
    return out.contiguous().view(B, out.shape[1] // nin, nin)[:, :, cond_in:].contiguous().view(B, -1)


def flow_compute_ll(blocks, x, n, solver="CCParallel", context=None):
    """blocks: list of dicts {mW, mb, mm, iW, ib, scaling} of CPU tensors.  Follows the reference call structure:
    per block forward (MADE + integral) and compute_log_jac (MADE again + one integrand evaluation)."""
    quad = integrate_parallel if solver == "CCParallel" else integrate_sequential
    d = x.shape[1]
    cond_in = context.shape[1] if context is not None else 0
    log_jac = 0.
    for blk in blocks:
        h = _embedding(blk, x, context, cond_in)
        z0 = h.view(h.shape[0], -1, d)[:, 0, :]
        z = torch.exp(blk["scaling"]).unsqueeze(0) * (quad(blk["iW"], blk["ib"], torch.zeros_like(x), x, h, n) + z0)
        h2 = _embedding(blk, x, context, cond_in)
        log_jac = log_jac + torch.log(integrand(blk["iW"], blk["ib"], x, h2, d) + 1e-10) + blk["scaling"].unsqueeze(0)
        x = torch.flip(z, [1])
    z = torch.flip(x, [1])
    return log_jac.sum(1) - .5 * (math.log(2 * math.pi) + z ** 2).sum(1), z


def blocks_from_state_dict(sd, nb_flow):
    def seq(prefix):
        idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
        return ([sd[f"{prefix}{j}.weight"] for j in idx], [sd[f"{prefix}{j}.bias"] for j in idx],
                [sd.get(f"{prefix}{j}.mask") for j in idx])
    out = []
    for i in range(nb_flow):
        mW, mb, mm = seq(f"Flow{i}.net.made.net.")
        iW, ib, _ = seq(f"Flow{i}.net.parallel_nets.net.")
        out.append(dict(mW=mW, mb=mb, mm=mm, iW=iW, ib=ib, scaling=sd[f"Flow{i}.scaling"]))
    return out
