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
