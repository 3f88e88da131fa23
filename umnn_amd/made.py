"""MADE conditioner (adjacent row a13 of the scope table): produces the embedding h the quadrature consumes.

Mirrors models/UMNN/made.py (MaskedLinear :16-27, MADE :30-144, ConditionnalMADE :146-195): constructor
signatures, ``net.{0,2,..}.{weight,bias,mask}`` state_dict keys and mask construction are the reference's.  The
GEMMs stay on PyTorch-ROCm (hipBLASLt): at the BSDS300 shape they are ~2 % of a block's FLOPs.
MI355X-side changes that do not alter results: the masked weight ``mask*W`` is cached between calls while the
weight is unchanged and no graph is being recorded, and ConditionnalMADE only computes the output columns it keeps.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class MaskedLinear(nn.Linear):
    """nn.Linear whose weight is multiplied elementwise by a fixed 0/1 ``mask`` buffer."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self.register_buffer('mask', torch.ones(out_features, in_features))
        self._cache = None      # (weight version, weight data_ptr, masked weight)

    def set_mask(self, mask):
        # mask arrives [in, out] (numpy, bool); stored [out, in] like the weight
        self.mask.data.copy_(torch.from_numpy(np.ascontiguousarray(mask.T).astype(np.float32)))
        self._cache = None

    def masked_weight(self):
        if torch.is_grad_enabled() and self.weight.requires_grad:
            return self.mask * self.weight
        key = (self.weight._version, self.weight.data_ptr(), self.mask._version, self.mask.data_ptr())
        if self._cache is None or self._cache[0] != key:
            self._cache = (key, (self.mask * self.weight).detach())
        return self._cache[1]

    def forward(self, input):
        return F.linear(input, self.masked_weight(), self.bias)


class MADE(nn.Module):
    def __init__(self, nin, hidden_sizes, nout, num_masks=1, natural_ordering=False, random=False, device="cpu"):
        super().__init__()
        assert nout % nin == 0, "nout must be integer multiple of nin"
        self.random, self.nin, self.nout, self.device = random, nin, nout, device
        self.pi = torch.tensor(math.pi).to(device)
        self.hidden_sizes = hidden_sizes
        sizes = [nin] + list(hidden_sizes) + [nout]
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(MaskedLinear(sizes[i], sizes[i + 1]))
            if i < len(sizes) - 2:
                layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers).to(device)
        self.natural_ordering, self.num_masks, self.seed = natural_ordering, num_masks, 0
        self.m = {}
        self.update_masks()

    def _degrees(self):
        """Degree vectors m[-1] (inputs) and m[l] (hidden layer l)."""
        L = len(self.hidden_sizes)
        rng = np.random.RandomState(self.seed)
        self.seed = (self.seed + 1) % self.num_masks
        deg = {}
        if self.random:
            deg[-1] = np.arange(self.nin) if self.natural_ordering else rng.permutation(self.nin)
            for l in range(L):
                deg[l] = rng.randint(deg[l - 1].min(), self.nin - 1, size=self.hidden_sizes[l])
        else:
            deg[-1] = np.arange(self.nin)
            for l in range(L):
                deg[l] = (self.nin - 1) - (np.arange(self.hidden_sizes[l]) % self.nin)
        return deg

    def update_masks(self):
        if self.m and self.num_masks == 1:
            return
        L = len(self.hidden_sizes)
        self.m = self._degrees()
        masks = [self.m[l - 1][:, None] <= self.m[l][None, :] for l in range(L)]     # hidden: non-strict
        masks.append(self.m[L - 1][:, None] < self.m[-1][None, :])                   # output: strict
        if self.nout > self.nin:
            masks[-1] = np.tile(masks[-1], (1, self.nout // self.nin))
        for layer, mask in zip((l for l in self.net if isinstance(l, MaskedLinear)), masks):
            layer.set_mask(mask)
        self.i_map = np.argsort(self.m[-1])

    def raw(self, x):
        """The masked MLP itself (what the flow's EmbeddingNetwork needs, whatever nout is)."""
        return self.net(x)

    def forward(self, x, context=None):
        if self.nout == 2:       # reference quirk (made.py:114-118): nout == 2 means "Gaussian MADE"
            out = self.net(x)
            mu, sigma = out[:, :self.nin], out[:, self.nin:]
            return (x - mu) * torch.exp(-sigma)
        return self.net(x)

    def compute_ll(self, x):
        out = self.net(x)
        mu, sigma = out[:, :self.nin], out[:, self.nin:]
        z = (x - mu) * torch.exp(-sigma)
        log_prob_gauss = -.5 * (torch.log(self.pi * 2) + z ** 2).sum(1)
        return -sigma.sum(1) + log_prob_gauss, z

    def invert(self, z):
        if self.nin != self.nout / 2:
            return None
        u = torch.zeros(z.shape)
        for d in range(self.nin):
            out = self.net(u)
            j = self.i_map[d]
            u[:, j] = z[:, j] * torch.exp(out[:, self.nin + j]) + out[:, j]
        return u


class ConditionnalMADE(MADE):
    """MADE over [context, x]; the context columns of every output chunk are dropped (made.py:165-168)."""

    def __init__(self, nin, cond_in, hidden_sizes, nout, num_masks=1, natural_ordering=False, random=False,
                 device="cpu"):
        super().__init__(nin + cond_in, hidden_sizes, nout, num_masks, natural_ordering, random, device)
        self.nin_non_cond, self.cond_in = nin, cond_in
        self._keep = None

    def _kept_rows(self, device):
        """Row indices of the last layer that survive the [:, :, cond_in:] slice, in output order."""
        if self._keep is None or self._keep.device != device:
            k = self.nout // self.nin
            idx = (torch.arange(k).view(-1, 1) * self.nin + torch.arange(self.cond_in, self.nin).view(1, -1))
            self._keep = idx.reshape(-1).to(device)
        return self._keep

    def raw(self, x, context):
        a = torch.cat((context, x), 1)
        layers = list(self.net)
        for layer in layers[:-1]:
            a = layer(a)
        last = layers[-1]
        keep = self._kept_rows(a.device)
        return F.linear(a, last.masked_weight().index_select(0, keep), last.bias.index_select(0, keep))

    def forward(self, x, context):
        if self.nout == 2:
            out = super().forward(torch.cat((context, x), 1))
            B = x.shape[0]
            return out.contiguous().view(B, out.shape[1] // self.nin, self.nin)[:, :, self.cond_in:] \
                .contiguous().view(B, -1)
        return self.raw(x, context)

    def computeLL(self, x, context):
        out = self.raw(x, context)
        n = self.nin_non_cond
        mu, sigma = out[:, :n], out[:, n:]
        z = (x - mu) * torch.exp(-sigma)
        log_prob_gauss = -.5 * (torch.log(self.pi * 2) + z ** 2).sum(1)
        return -sigma.sum(1) + log_prob_gauss, z
