"""Clenshaw-Curtis tables (host logic).

Mirrors ``compute_cc_weights`` of the reference (models/UMNN/ParallelNeuralIntegral.py:14-34, duplicated at
NeuralIntegral.py:14-34 and UMNNMAF.py:55-69): same name, same return value
``(cc_weights[n+1,1], steps[n+1,1])`` as fp32 CPU tensors, same module-level cache.  The device copies the
HIP kernels read are cached per (n, device) so no host->device transfer happens on the hot path.
"""
import math

import numpy as np
import torch

_cc_weights_cache = {}
_device_cache = {}


def _tables_f64(nb_steps):
    """w = Lambda^T W in float64.  Lambda_jk = cos(jk*pi/n)*2/n with column 0 -> 1/n and column n halved;
    W_j = 2/(1-j^2) on even j (W_0 = 1), 0 on odd j.  Written with the node index leading so that the
    product is one matrix-vector contraction."""
    n = int(nb_steps)
    if n < 1:
        raise ValueError("nb_steps must be >= 1")
    idx = np.arange(n + 1)
    lam_t = np.cos(np.outer(idx, idx) * math.pi / n)       # [k, j] (symmetric before the edits)
    lam_t[0, :] = .5
    lam_t[-1, :] = .5 * lam_t[-1, :]
    lam_t = lam_t * 2 / n
    W = np.zeros(n + 1)
    even = idx[::2]
    W[even] = 2. / (1. - even.astype(np.float64) ** 2)
    W[0] = 1.
    return lam_t @ W, np.cos(idx * math.pi / n)


def compute_cc_weights(nb_steps):
    key = int(nb_steps)
    hit = _cc_weights_cache.get(key)
    if hit is None:
        w, s = _tables_f64(key)
        hit = (torch.from_numpy(w.reshape(-1, 1)).float(), torch.from_numpy(s.reshape(-1, 1)).float())
        _cc_weights_cache[key] = hit
    return hit


def device_tables(nb_steps, device):
    """Flat fp32 (w, s) on ``device``; uploaded once per (n, device)."""
    device = torch.device(device)
    key = (int(nb_steps), device.type, device.index)
    hit = _device_cache.get(key)
    if hit is None:
        w, s = compute_cc_weights(nb_steps)
        hit = (w.reshape(-1).contiguous().to(device), s.reshape(-1).contiguous().to(device))
        _device_cache[key] = hit
    return hit
