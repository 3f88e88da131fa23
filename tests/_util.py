"""Shared helpers for the parity tests: load golden fixtures, build oracle objects from them."""
import glob
import os

import numpy as np

from oracle import cc_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def g2_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g2_*.npz")))


def g7_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g7_*.npz")))


def g4_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g4_*.npz")))


def net_from_g2(G, dtype=np.float32):
    L = len(G["hidden"]) + 1
    out_act = O.SIGMOID if str(G["act"]) == "Sigmoid" else O.ELU1
    return O.Net([G[f"W{l}"].astype(dtype) for l in range(L)], [G[f"b{l}"].astype(dtype) for l in range(L)],
                 O.LEAKY, out_act)


def _seq(sd, prefix, dtype):
    """Collect Linear layers of an nn.Sequential stored under prefix ('...net.') in order."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
    Ws = [sd[f"{prefix}{i}.weight"].astype(dtype) for i in idx]
    bs = [sd[f"{prefix}{i}.bias"].astype(dtype) for i in idx]
    masks = [sd.get(f"{prefix}{i}.mask") for i in idx]
    return Ws, bs, masks


def state_dict_of(G):
    return {k[3:]: v for k, v in G.items() if k.startswith("sd/")}


def blocks_from_g4(G, dtype=np.float32):
    sd = state_dict_of(G)
    blocks = []
    for i in range(int(G["nb_flow"])):
        mW, mb, mm = _seq(sd, f"Flow{i}.net.made.net.", dtype)
        iW, ib, _ = _seq(sd, f"Flow{i}.net.parallel_nets.net.", dtype)
        blocks.append(O.Block(mW, mb, mm, O.Net(iW, ib, O.LEAKY, O.ELU1),
                              sd[f"Flow{i}.scaling"].astype(dtype), int(G["cond_in"])))
    return blocks


def rel_err(a, b):
    """max |a-b| / max(|b|, 1): the SURVEY 8(d) criterion."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) if a.size else 0.0


def scaled_err(a, b):
    """max |a-b| / max|b| -- for gradient tensors whose entries span many magnitudes."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)) if a.size else 0.0


def kink_margin(net, x0, x, h, nb_steps):
    """Smallest |pre-activation| of any hidden unit at any quadrature node, RELATIVE to the sum of the magnitudes of the
    terms that form it (what rounding error scales with).  The backward multiplies by act'(z) = 1 or slope: where |z| is
    below the rounding noise of the dot product (~sqrt(K) 2^-24 of sum|terms|, a few 1e-7 -- seed 19 of the random sweep has margin 2e-8 and the fp32 and fp64 ORACLES already differ by 7e-4 on it), the sign -- and with it one
    unit's whole contribution at that point -- is decided by summation order, in the reference as much as here.  A case
    whose margin is above the noise must meet the strict tolerance; a case below it is kink-ambiguous."""
    x0, x, h = (np.asarray(a, np.float64) for a in (x0, x, h))
    net64 = O.Net([W.astype(np.float64) for W in net.Ws], [b.astype(np.float64) for b in net.bs], net.hidden_act, net.out_act)
    w, s = O.cc_tables(nb_steps, np.float64)
    B, d = x.shape
    t, _ = O._nodes(x0, x, s, nb_steps)
    n1 = nb_steps + 1
    hs = np.broadcast_to(h[:, None, :], (B, n1, h.shape[1])).reshape(B * n1, -1)
    a = O.rows_from(t.reshape(B * n1, d), hs, d)
    margin = np.inf
    for l in range(len(net64.Ws) - 1):
        W, b = net64.Ws[l], net64.bs[l]
        z = a @ W.T + b
        mag = np.abs(a) @ np.abs(W).T + np.abs(b)
        margin = min(margin, float(np.min(np.abs(z) / np.maximum(mag, 1e-300))))
        a = O._hidden(z, net64.hidden_act)
    return margin


def kink_margin_rows(net, x0, x, h, nb_steps):
    """kink_margin per SAMPLE ROW: [B] -- the smallest |pre-activation| / sum|terms| over every hidden unit, dimension and
    quadrature node of that row (float64).  The per-row backward outputs (d_h, d_x with a g_fx cotangent) depend on their own
    row only, so a row whose margin is above the rounding noise of the arithmetic under test must agree with the reference; a
    row below it may legitimately sit on either side of a LeakyReLU kink -- in the reference's own float32 run as much as here."""
    x0, x, h = (np.asarray(a, np.float64) for a in (x0, x, h))
    net64 = O.Net([W.astype(np.float64) for W in net.Ws], [b.astype(np.float64) for b in net.bs], net.hidden_act, net.out_act)
    w, s = O.cc_tables(nb_steps, np.float64)
    B, d = x.shape
    t, _ = O._nodes(x0, x, s, nb_steps)
    n1 = nb_steps + 1
    hs = np.broadcast_to(h[:, None, :], (B, n1, h.shape[1])).reshape(B * n1, -1)
    a = O.rows_from(t.reshape(B * n1, d), hs, d)                 # rows ordered (b, k, i)
    margin = np.full(B, np.inf)
    for l in range(len(net64.Ws) - 1):
        W, b = net64.Ws[l], net64.bs[l]
        z = a @ W.T + b
        mag = np.abs(a) @ np.abs(W).T + np.abs(b)
        m = (np.abs(z) / np.maximum(mag, 1e-300)).reshape(B, -1).min(axis=1)
        margin = np.minimum(margin, m)
        a = O._hidden(z, net64.hidden_act)
    return margin
