"""CPU oracle for the UMNN neural-integration hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm, written from the
math; it is the *checker* for the HIP path, never the product.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``umnn_amd/`` imports it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against golden vectors produced by importing the reference itself
(``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` is mounted) and against the reference's own analytic
known-answer tests (``tests/test_numerical_validation.py:18-97,319-402``).

Reference lines restated (all relative to /root/reference):
  cc_tables              models/UMNN/ParallelNeuralIntegral.py:14-34
  mlp_rows / integrand   models/UMNN/UMNNMAF.py:263-284, MonotonicNN.py:12-27
  integrate_parallel     models/UMNN/ParallelNeuralIntegral.py:37-65
  integrate_sequential   models/UMNN/NeuralIntegral.py:37-66
  integrate_backward     models/UMNN/ParallelNeuralIntegral.py:66-94,110-123
  made_forward           models/UMNN/made.py:16-27,74-119,146-168
  block_* / flow_*       models/UMNN/UMNNMAF.py:76-162, UMNNMAFFlow.py:72-137
  monotonic_forward      models/UMNN/MonotonicNN.py:49-54

All arithmetic is carried out in ``dtype`` (float32 to mirror the reference,
float64 for a high-precision cross-check).
"""
import math

import numpy as np

LEAKY, RELU = 0, 1          # hidden activations (nn.LeakyReLU(0.01) / nn.ReLU)
ELU1, SIGMOID = 0, 1        # output activations (ELU(.)+1 / Sigmoid)


# --------------------------------------------------------------------------
# Clenshaw-Curtis tables
# --------------------------------------------------------------------------
def cc_tables(nb_steps, dtype=np.float32):
    """Nodes s_k = cos(k pi / n) and weights w = Lambda^T W (float64 -> dtype).

    Restates ParallelNeuralIntegral.py:19-30: Lambda_jk = cos(jk pi/n) with
    column 0 := 1/2 and column n halved, scaled by 2/n; W_j = 2/(1-j^2) for
    even j, W_0 = 1, 0 for odd j.  (The reference builds W in an *integer*
    array, so 2/(1-j^2) is true division on ints -> float64.)
    """
    n = int(nb_steps)
    k = np.arange(0, n + 1, 1).reshape(-1, 1)
    lam = np.cos((k @ k.T) * math.pi / n)
    lam[:, 0] = .5
    lam[:, -1] = .5 * lam[:, -1]
    lam = lam * 2 / n
    j = np.arange(0, n + 1, 1).reshape(-1, 1)
    j[np.arange(1, n + 1, 2)] = 0
    W = 2 / (1 - j ** 2)
    W[0] = 1
    W[np.arange(1, n + 1, 2)] = 0
    w = (lam.T @ W).astype(dtype).reshape(-1)
    s = np.cos(np.arange(0, n + 1, 1) * math.pi / n).astype(dtype)
    return w, s


# --------------------------------------------------------------------------
# Integrand MLP
# --------------------------------------------------------------------------
class Net:
    """Plain container: Ws[l] is [out,in] (torch.nn.Linear layout), bs[l] is [out]."""

    def __init__(self, Ws, bs, hidden_act=LEAKY, out_act=ELU1):
        self.Ws = [np.asarray(W) for W in Ws]
        self.bs = [np.asarray(b) for b in bs]
        self.hidden_act = hidden_act
        self.out_act = out_act

    def astype(self, dtype):
        return Net([W.astype(dtype) for W in self.Ws], [b.astype(dtype) for b in self.bs],
                   self.hidden_act, self.out_act)

    @property
    def n_params(self):
        return sum(W.size + b.size for W, b in zip(self.Ws, self.bs))


def _hidden(z, act):
    if act == LEAKY:
        return np.where(z > 0, z, z * z.dtype.type(0.01))
    return np.maximum(z, z.dtype.type(0))


def _hidden_grad(z, act):
    if act == LEAKY:
        return np.where(z > 0, z.dtype.type(1), z.dtype.type(0.01))
    return (z > 0).astype(z.dtype)


def _out(z, act):
    one = z.dtype.type(1)
    if act == ELU1:
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0))) + one
    return one / (one + np.exp(-z))


def _out_grad(z, act):
    one = z.dtype.type(1)
    if act == ELU1:
        return np.where(z > 0, one, np.exp(np.minimum(z, 0)))
    s = one / (one + np.exp(-z))
    return s * (one - s)


def rows_from(x, h, d):
    """[B,d],[B,E*d] -> [B*d, 1+E]: row (b,i) = [x_bi, h_b[0*d+i], ..., h_b[(E-1)*d+i]].

    Restates the cat/view/transpose of UMNNMAF.py:265,279-281 (feature-major,
    dim-minor embedding layout) and the plain cat of MonotonicNN.py:27 (d=1).
    """
    B = x.shape[0]
    cat = np.concatenate([x, h], axis=1)
    return cat.reshape(B, -1, d).transpose(0, 2, 1).reshape(B * d, -1)


def mlp_rows(net, rows, keep=False):
    """Apply the shared MLP to rows [R, 1+E] -> [R].  keep=True also returns pre-activations."""
    a = rows
    pres, acts = [], [rows]
    L = len(net.Ws)
    for l in range(L):
        z = a @ net.Ws[l].T + net.bs[l]
        pres.append(z)
        a = _hidden(z, net.hidden_act) if l < L - 1 else _out(z, net.out_act)
        if l < L - 1:
            acts.append(a)
    out = a[:, 0]
    return (out, pres, acts) if keep else out


def integrand(net, x, h):
    """f(x;h): [B,d],[B,E*d] -> [B,d]."""
    B, d = x.shape
    return mlp_rows(net, rows_from(x, h, d)).reshape(B, d)


# --------------------------------------------------------------------------
# Quadrature
# --------------------------------------------------------------------------
def _nodes(x0, x, s, nb_steps):
    """t[b,k,i] exactly as the reference forms it (ParallelNeuralIntegral.py:49-55):
    xT = x0 + n*((x-x0)/n);  t = x0 + (xT-x0)*(s+1)/2."""
    dt = x.dtype.type
    step = (x - x0) / dt(nb_steps)
    xT = x0 + dt(nb_steps) * step
    t = x0[:, None, :] + (xT - x0)[:, None, :] * (s[None, :, None] + dt(1)) / dt(2)
    return t, xT


def integrate_parallel(net, x0, x, h, nb_steps, inv_f=False):
    """All n+1 nodes as one batch; F = (xT-x0)/2 * sum_k w_k f(t_k).  -> [B,d]"""
    dtype = x.dtype
    w, s = cc_tables(nb_steps, dtype)
    B, d = x.shape
    t, xT = _nodes(x0, x, s, nb_steps)
    n1 = nb_steps + 1
    hs = np.broadcast_to(h[:, None, :], (B, n1, h.shape[1])).reshape(B * n1, -1)
    f = integrand(net, t.reshape(B * n1, d), hs).reshape(B, n1, d)
    if inv_f:
        f = dtype.type(1) / f
    return (f * w[None, :, None]).sum(1) * (xT - x0) / dtype.type(2)


def integrate_sequential(net, x0, x, h, nb_steps):
    """Loop over nodes with a running sum (NeuralIntegral.py:53-61)."""
    dtype = x.dtype
    w, s = cc_tables(nb_steps, dtype)
    dt = dtype.type
    step = (x - x0) / dt(nb_steps)
    xT = x0 + dt(nb_steps) * step
    z = np.zeros_like(x)
    for k in range(nb_steps + 1):
        t = x0 + (xT - x0) * (s[k] + dt(1)) / dt(2)
        z = z + w[k] * integrand(net, t, h)
    return z * (xT - x0) / dt(2)


def integrate_backward(net, x0, x, h, nb_steps, g, inv_f=False):
    """Gradients the reference's custom backward returns for cotangent g [B,d].

    inv_f (ParallelNeuralIntegral.py:70-72): d_theta and d_h differentiate 1/f, i.e. the node cotangent is
    multiplied by -1/f^2; the Leibniz terms keep f itself (both branches of :120-123 are identical).

    d_theta, d_h: VJP of f at every node with cotangent g*(xT-x0)/2*w_k
    (ParallelNeuralIntegral.py:70-71,91-94).  d_x = f(x;h)*g, d_x0 = -f(x0;h)*g
    (Leibniz, :117-123) -- NOT the derivative of the discrete quadrature.
    Returns dx0, dx, dh, dWs(list), dbs(list), and the flat dtheta in
    ``parameters()`` order (W0,b0,W1,b1,...).
    """
    dtype = x.dtype
    dt = dtype.type
    w, s = cc_tables(nb_steps, dtype)
    B, d = x.shape
    E = h.shape[1] // d
    n1 = nb_steps + 1
    t, xT = _nodes(x0, x, s, nb_steps)
    cot = (g * (xT - x0) / dt(2))[:, None, :] * w[None, :, None]          # [B,n1,d]
    hs = np.broadcast_to(h[:, None, :], (B, n1, h.shape[1])).reshape(B * n1, -1)
    rows = rows_from(t.reshape(B * n1, d), hs, d)                         # [B*n1*d, 1+E]
    _, pres, acts = mlp_rows(net, rows, keep=True)
    L = len(net.Ws)
    dout = cot.reshape(-1) * _out_grad(pres[-1][:, 0], net.out_act)
    if inv_f:
        fval = _out(pres[-1][:, 0], net.out_act)
        dout = -dout / (fval * fval)
    delta = dout[:, None]                                                 # [R,1]
    dWs, dbs = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        dWs[l] = delta.T @ acts[l]
        dbs[l] = delta.sum(0)
        if l > 0:
            delta = (delta @ net.Ws[l]) * _hidden_grad(pres[l - 1], net.hidden_act)
    d_rows = delta @ net.Ws[0]                                            # [R, 1+E]
    # rows are ordered (b, k, i); columns 1.. are h_e -> scatter back to [B, E*d] layout e*d+i
    dh = d_rows[:, 1:].reshape(B, n1, d, E).sum(1).transpose(0, 2, 1).reshape(B, E * d)
    dx = integrand(net, x, h) * g
    dx0 = -integrand(net, x0, h) * g
    flat = np.concatenate([np.concatenate([dW.reshape(-1), db.reshape(-1)]) for dW, db in zip(dWs, dbs)])
    return dx0, dx, dh, dWs, dbs, flat


# --------------------------------------------------------------------------
# MADE conditioner (adjacent row a13) and flow blocks
# --------------------------------------------------------------------------
def made_masks(nin, hidden_sizes, nout):
    """Natural-ordering masks (made.py:85-100): degrees nin-1-(i mod nin); '<=' hidden, '<' output,
    output mask tiled nout/nin times.  Returned in [out,in] (Linear) layout."""
    m = {-1: np.arange(nin)}
    L = len(hidden_sizes)
    for l in range(L):
        m[l] = np.array([nin - 1 - (i % nin) for i in range(hidden_sizes[l])])
    masks = [m[l - 1][:, None] <= m[l][None, :] for l in range(L)]
    masks.append(m[L - 1][:, None] < m[-1][None, :])
    if nout > nin:
        masks[-1] = np.concatenate([masks[-1]] * int(nout / nin), axis=1)
    return [mk.astype(np.uint8).T for mk in masks]


def made_forward(Ws, bs, masks, x):
    """ReLU MLP with masked weights (made.py:27,113-119; the nout==2 Gaussian branch is not on the path)."""
    a = x
    for l, (W, b, mk) in enumerate(zip(Ws, bs, masks)):
        a = a @ (mk.astype(W.dtype) * W).T + b
        if l < len(Ws) - 1:
            a = np.maximum(a, a.dtype.type(0))
    return a


def cond_made_forward(Ws, bs, masks, x, context, nin_total, cond_in):
    """ConditionnalMADE.forward (made.py:165-168): context is prepended, its output columns dropped."""
    out = made_forward(Ws, bs, masks, np.concatenate([context, x], axis=1))
    B = x.shape[0]
    return np.ascontiguousarray(out.reshape(B, out.shape[1] // nin_total, nin_total)[:, :, cond_in:]).reshape(B, -1)


class Block:
    """One UMNNMAF block: MADE params + integrand Net + frozen scaling."""

    def __init__(self, made_Ws, made_bs, made_masks_, net, scaling, cond_in=0):
        self.made_Ws, self.made_bs, self.made_masks = made_Ws, made_bs, made_masks_
        self.net, self.scaling, self.cond_in = net, scaling, cond_in

    def embed(self, x, context=None):
        if self.cond_in > 0:
            return cond_made_forward(self.made_Ws, self.made_bs, self.made_masks, x, context,
                                     x.shape[1] + self.cond_in, self.cond_in)
        return made_forward(self.made_Ws, self.made_bs, self.made_masks, x)


def block_forward(blk, x, nb_steps, solver="CCParallel", context=None):
    """z = exp(scaling) * (int_0^x f + h[:,0,:])   (UMNNMAF.py:76-134)."""
    h = blk.embed(x, context)
    d = x.shape[1]
    z0 = h.reshape(h.shape[0], -1, d)[:, 0, :]
    x0 = np.zeros_like(x)
    if solver == "CC":
        F = integrate_sequential(blk.net, x0, x, h, nb_steps)
    else:
        F = integrate_parallel(blk.net, x0, x, h, nb_steps)
    return np.exp(blk.scaling)[None, :] * (F + z0), h


def block_log_jac(blk, x, h):
    """log(f(x;h)+1e-10) + scaling   (UMNNMAF.py:136-139)."""
    return np.log(integrand(blk.net, x, h) + x.dtype.type(1e-10)) + blk.scaling[None, :]


def flow_compute_ll(blocks, x, nb_steps, solver="CCParallel", context=None):
    """UMNNMAFFlow.compute_ll (UMNNMAFFlow.py:109-119) -> (ll [B], z [B,d])."""
    dt = x.dtype.type
    log_jac = 0.
    for blk in blocks:
        z, h = block_forward(blk, x, nb_steps, solver, context)
        log_jac = log_jac + block_log_jac(blk, x, h)
        x = z[:, ::-1]
    z = x[:, ::-1]
    log_prob_gauss = dt(-.5) * (np.log(dt(math.pi) * dt(2)) + z ** 2).sum(1)
    return log_jac.sum(1) + log_prob_gauss, z


def flow_forward(blocks, x, nb_steps, solver="CCParallel", context=None):
    for blk in blocks:
        z, _ = block_forward(blk, x, nb_steps, solver, context)
        x = z[:, ::-1]
    return x[:, ::-1]


def flow_log_jac(blocks, x, nb_steps, solver="CCParallel", context=None):
    """UMNNMAFFlow.compute_log_jac_bis (UMNNMAFFlow.py:100-107) -> (z, log_jac [B,d])."""
    log_jac = 0.
    for blk in blocks:
        z, h = block_forward(blk, x, nb_steps, solver, context)
        log_jac = log_jac + block_log_jac(blk, x, h)
        x = z[:, ::-1]
    return x[:, ::-1], log_jac


def monotonic_forward(integrand_net, cond_Ws, cond_bs, x, h, nb_steps):
    """MonotonicNN.forward (MonotonicNN.py:49-54): exp(s(h)) * int_0^x f(t,h) dt + o(h)."""
    a = h
    for l, (W, b) in enumerate(zip(cond_Ws, cond_bs)):
        a = a @ W.T + b
        if l < len(cond_Ws) - 1:
            a = np.maximum(a, a.dtype.type(0))
    offset, scaling = a[:, [0]], np.exp(a[:, [1]])
    F = integrate_parallel(integrand_net, np.zeros_like(x), x, h, nb_steps)
    return scaling * F + offset


def integrand_vjp(net, x, h, cot):
    """VJP of f(x;h) [B,d] with cotangent ``cot``: what plain autograd through IntegrandNetwork.forward gives the
    reference for the log-det term (UMNNMAF.py:138,143,148).  -> (dx, dh, flat dtheta)."""
    B, d = x.shape
    E = h.shape[1] // d
    rows = rows_from(x, h, d)
    _, pres, acts = mlp_rows(net, rows, keep=True)
    L = len(net.Ws)
    delta = (cot.reshape(-1) * _out_grad(pres[-1][:, 0], net.out_act))[:, None]
    dWs, dbs = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        dWs[l] = delta.T @ acts[l]
        dbs[l] = delta.sum(0)
        if l > 0:
            delta = (delta @ net.Ws[l]) * _hidden_grad(pres[l - 1], net.hidden_act)
    d_rows = delta @ net.Ws[0]                       # [B*d, 1+E], rows ordered (b, i)
    dx = d_rows[:, 0].reshape(B, d)
    dh = d_rows[:, 1:].reshape(B, d, E).transpose(0, 2, 1).reshape(B, E * d)
    flat = np.concatenate([np.concatenate([dW.reshape(-1), db.reshape(-1)]) for dW, db in zip(dWs, dbs)])
    return dx, dh, flat
