"""Integrand networks and the MLP description the HIP kernels consume.

``IntegrandNetwork`` mirrors models/UMNN/UMNNMAF.py:235-301 (one MLP shared by all d dimensions, applied to the
rows [x_i, h_{0,i}, ..., h_{E-1,i}]); ``IntegrandNN`` mirrors models/UMNN/MonotonicNN.py:12-27.  Constructor
signatures, attribute names and ``state_dict`` keys (``net.{0,2,...}.{weight,bias}``) are the reference's.

``mlp_spec(integrand)`` recognises an integrand whose arithmetic is "Linear -> (LeakyReLU(0.01)|ReLU) -> ... ->
Linear -> (ELU+1|Sigmoid)" and returns what the C ABI needs; anything else (lambdas, custom modules) yields None
and is integrated by the generic ATen quadrature in integral.py.
"""
from collections import namedtuple

import torch
import torch.nn as nn

from . import _lib

MlpSpec = namedtuple("MlpSpec", "linears hidden_act out_act")


class ELUPlus(nn.Module):
    """ELU(x) + 1  (UMNNMAF.py:11-16); the submodule name ``elu`` is part of the reference's module tree."""

    def __init__(self):
        super().__init__()
        self.elu = nn.ELU()

    def forward(self, x):
        return self.elu(x) + 1.


def _output_activation(name):
    if name == "ELU":
        return ELUPlus()
    if name == "Sigmoid":
        return nn.Sigmoid()
    raise KeyError(name)       # the reference indexes a dict: unknown names raise KeyError


def compute_lipschitz_linear(W, nb_iter=10):
    """Spectral-norm estimate by power iteration on W^T W (UMNNMAF.py:26-34)."""
    v = torch.randn(W.shape[1], 1, device=W.device, dtype=W.dtype)
    for _ in range(nb_iter):
        v = W.t() @ (W @ v)
        v = v / torch.norm(v)
    return (torch.norm(W.t() @ (W @ v)) / torch.norm(v)) ** .5


class IntegrandNetwork(nn.Module):
    def __init__(self, nnets, nin, hidden_sizes, nout, act_func='ELU', device="cpu"):
        super().__init__()
        self.nin, self.nnets, self.nout = nin, nnets, nout
        self.hidden_sizes = hidden_sizes
        self.device = device
        sizes = [nin] + list(hidden_sizes) + [nout]
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[i], sizes[i + 1]))
            layers.append(nn.LeakyReLU() if i < len(sizes) - 2 else _output_activation(act_func))
        self.net = nn.Sequential(*layers)
        self.masks = torch.eye(nnets).to(device)      # unused by the arithmetic; kept for attribute parity

    def to(self, device):
        self.device = device
        self.net.to(device)
        self.masks = self.masks.to(device)
        return self

    def rows(self, x, h):
        """[B,d],[B,E*d] -> [B*d, 1+E]; h is feature-major / dim-minor (index e*d+i)."""
        B = x.shape[0]
        stacked = torch.cat((x, h), 1).view(B, -1, self.nnets)      # [B, 1+E, d]
        return stacked.transpose(1, 2).reshape(B * self.nnets, -1)

    def forward(self, x, h):
        return self.net(self.rows(x, h)).view(x.shape[0], -1)

    def independant_forward(self, x):
        return self.net(x)

    def compute_lipschitz(self, nb_iter=10):
        with torch.no_grad():
            L = 1
            for layer in self.net:
                if isinstance(layer, nn.Linear):
                    L = L * compute_lipschitz_linear(layer.weight, nb_iter)
        return L

    def force_lipschitz(self, L=1.5):
        with torch.no_grad():
            for layer in self.net:
                if isinstance(layer, nn.Linear):
                    layer.weight /= max(compute_lipschitz_linear(layer.weight, 10) / L, 1)

    # names the reference's callers use but the reference never defines: aliases, a superset API
    computeLipshitz = compute_lipschitz
    forceLipshitz = force_lipschitz

    def _umnn_spec(self):
        return _spec_from_sequential(self.net, plus_one_outside=False)


class IntegrandNN(nn.Module):
    def __init__(self, in_d, hidden_layers):
        super().__init__()
        sizes = [in_d] + list(hidden_layers) + [1]
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[i], sizes[i + 1]))
            layers.append(nn.ReLU() if i < len(sizes) - 2 else nn.ELU())
        self.net = nn.Sequential(*layers)

    def forward(self, x, h):
        return self.net(torch.cat((x, h), 1)) + 1.

    def _umnn_spec(self):
        return _spec_from_sequential(self.net, plus_one_outside=True)


def _spec_from_sequential(seq, plus_one_outside):
    mods = list(seq)
    if len(mods) < 4 or len(mods) % 2:
        return None
    linears, hidden = [], None
    for i in range(0, len(mods), 2):
        lin, act = mods[i], mods[i + 1]
        if type(lin) is not nn.Linear or lin.bias is None or lin.weight.dtype != torch.float32:
            return None
        linears.append(lin)
        if i + 2 < len(mods):
            if isinstance(act, nn.LeakyReLU) and abs(act.negative_slope - 0.01) < 1e-12:
                kind = _lib.ACT_LEAKY_RELU
            elif isinstance(act, nn.ReLU):
                kind = _lib.ACT_RELU
            else:
                return None
            if hidden is not None and hidden != kind:
                return None
            hidden = kind
        else:
            name = type(act).__name__
            if plus_one_outside:
                if not (isinstance(act, nn.ELU) and act.alpha == 1.0):
                    return None
                out = _lib.OUT_ELU_PLUS_ONE
            elif name == "ELUPlus" and isinstance(getattr(act, "elu", None), nn.ELU) and act.elu.alpha == 1.0:
                out = _lib.OUT_ELU_PLUS_ONE
            elif isinstance(act, nn.Sigmoid):
                out = _lib.OUT_SIGMOID
            else:
                return None
    if linears[-1].out_features != 1 or len(linears) > _lib.MAX_LINEAR:
        return None
    for a, b in zip(linears, linears[1:]):
        if a.out_features != b.in_features or a.out_features > 127:
            return None
    return MlpSpec(linears, hidden, out)


def mlp_spec(integrand):
    """MlpSpec for integrands the HIP kernels can evaluate, else None.

    Also recognises the reference's own classes by structure (same class names, same module tree), so a model
    built from reference code can be handed to ``ParallelNeuralIntegral.apply`` of this package unchanged."""
    if not isinstance(integrand, nn.Module):
        return None
    # cached on the module while its Sequential is the same object with the same children (the spec holds the Linear
    # modules themselves, so in-place weight updates and .to(device) are seen through it)
    net = getattr(integrand, "net", None)
    hit = integrand.__dict__.get("_umnn_spec_cache")
    if hit is not None and hit[0] is net and hit[1] == _spec_key(net):
        return hit[2]
    spec = _mlp_spec_uncached(integrand)
    if isinstance(net, nn.Sequential):
        integrand.__dict__["_umnn_spec_cache"] = (net, _spec_key(net), spec)
    return spec


def _spec_key(net):
    if not isinstance(net, nn.Sequential):
        return None
    mods = tuple(net._modules.values())
    w = getattr(mods[0], "weight", None) if mods else None
    return mods + ((w.dtype, w.device) if w is not None else ())


def _mlp_spec_uncached(integrand):
    fn = getattr(integrand, "_umnn_spec", None)
    if fn is not None:
        return fn()
    if not isinstance(getattr(integrand, "net", None), nn.Sequential):
        return None
    cls = type(integrand).__name__
    if cls == "IntegrandNetwork":
        return _spec_from_sequential(integrand.net, plus_one_outside=False)
    if cls == "IntegrandNN":
        return _spec_from_sequential(integrand.net, plus_one_outside=True)
    return None
