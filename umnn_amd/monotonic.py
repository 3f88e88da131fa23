"""MonotonicNN: a scalar function monotone in x, y = exp(s(h)) * int_0^x f(t;h) dt + o(h).

Mirrors models/UMNN/MonotonicNN.py:29-54 (constructor ``MonotonicNN(in_d, hidden_layers, nb_steps=50, dev="cpu")``,
``forward(x, h)`` with x [B,1] and h [B,in_d-1], state_dict keys ``integrand.net.*`` / ``net.*``).  The integral is
the same HIP kernel as the flow's (d = 1, E = in_d-1, ReLU hidden layers).
"""
import torch
import torch.nn as nn

from .integral import ParallelNeuralIntegral, _flatten
from .nets import IntegrandNN  # noqa: F401  (re-exported)


class MonotonicNN(nn.Module):
    def __init__(self, in_d, hidden_layers, nb_steps=50, dev="cpu"):
        super().__init__()
        self.integrand = IntegrandNN(in_d, hidden_layers)
        sizes = [in_d - 1] + list(hidden_layers) + [2]      # conditioner -> (offset, log-scale)
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[i], sizes[i + 1]))
            if i < len(sizes) - 2:
                layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers)
        self.device = dev
        self.nb_steps = nb_steps

    def forward(self, x, h):
        x0 = torch.zeros_like(x)
        out = self.net(h)
        offset, scaling = out[:, [0]], torch.exp(out[:, [1]])
        integral = ParallelNeuralIntegral.apply(x0, x, self.integrand, _flatten(self.integrand.parameters()), h,
                                                self.nb_steps)
        return scaling * integral + offset
