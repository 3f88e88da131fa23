"""UMNN-MAF flow modules: EmbeddingNetwork, UMNNMAF (one block) and UMNNMAFFlow (the stack).

The nn.Module API (constructor arguments, method names, ``.to`` returning self, state_dict keys
``Flow{i}.net.made.net.*``, ``Flow{i}.net.parallel_nets.net.*``, ``Flow{i}.{scaling,pi,cc_weights,cc_steps}``, ``pi``)
is the reference's: models/UMNN/UMNNMAF.py:37-232,304-329 and models/UMNN/UMNNMAFFlow.py:8-151.

What is different underneath (MI355X-first, results unchanged):
  * one block = ONE conditioner pass + ONE fused HIP launch giving both z and log|dz/dx| (quadrature node 0 is x,
    so f(x;h) falls out of the integral; the reference runs MADE twice and the integrand once more,
    UMNNMAFFlow.py:113-114 / UMNNMAF.py:136-139);
  * the node axis is never materialised, so "CC" and "CCParallel" are the same kernel;
  * training goes through ``IntegralWithJacobian`` (HIP forward + HIP backward with the reference's gradient
    convention); inference through the fully fused block epilogue.
Names the reference's scripts call but the reference never defines exist here as aliases.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import integral as _I
from .integral import NeuralIntegral, ParallelNeuralIntegral, IntegralWithJacobian, IntegralWithJacobianParams, _flatten  # noqa: F401
from .made import MADE, ConditionnalMADE
from .nets import ELUPlus, IntegrandNetwork, compute_lipschitz_linear, mlp_spec  # noqa: F401  (re-exported)
from .quadrature import compute_cc_weights

_SOLVERS = {"CC": NeuralIntegral, "CCParallel": ParallelNeuralIntegral}


class EmbeddingNetwork(nn.Module):
    """Conditioner (MADE) + the shared integrand MLP of one block."""

    def __init__(self, in_d, hiddens_embedding=[50, 50, 50, 50], hiddens_integrand=[50, 50, 50, 50], out_made=1,
                 cond_in=0, act_func='ELU', device="cpu"):
        super().__init__()
        self.m_embeding = None
        self.embedding_dtype = None     # None: whatever the conditioner computes in; torch.bfloat16: configuration C4
        self.device = device
        self.in_d = in_d
        if cond_in > 0:
            self.made = ConditionnalMADE(in_d, cond_in, hiddens_embedding, (in_d + cond_in) * out_made, num_masks=1,
                                         natural_ordering=True).to(device)
        else:
            self.made = MADE(in_d, hiddens_embedding, in_d * out_made, num_masks=1, natural_ordering=True).to(device)
        self.parallel_nets = IntegrandNetwork(in_d, 1 + out_made, hiddens_integrand, 1, act_func=act_func,
                                              device=device)

    def to(self, device):
        self.device = device
        self.made.to(device)
        self.parallel_nets.to(device)
        return self

    def make_embeding(self, x_made, context=None):
        # .raw(): the masked MLP itself.  (The reference calls MADE.forward, whose nout==2 "Gaussian" branch
        # breaks the d=2/E=1 and d=1/E=2 flows, made.py:114-118; the embedding never wants that branch.)
        if isinstance(self.made, ConditionnalMADE):
            self.m_embeding = self.made.raw(x_made, context, out_dtype=self.embedding_dtype)
        else:
            self.m_embeding = self.made.raw(x_made, out_dtype=self.embedding_dtype)
        return self.m_embeding

    def forward(self, x_t):
        return self.parallel_nets.forward(x_t, self.m_embeding)


class UMNNMAF(nn.Module):
    def __init__(self, net, input_size, nb_steps=100, device="cpu", solver="CC"):
        super().__init__()
        self.net = net.to(device)
        self.device = device
        self.input_size = input_size
        self.nb_steps = nb_steps
        self.solver = solver
        self.register_buffer("pi", torch.tensor(math.pi))
        # Registered tables keep the CONSTRUCTOR's shape for the whole life of the module, like the reference's
        # (state_dict parity: UCIExperiments.py:131-153 saves after set_steps_nb).  They are checkpoint payload only:
        # the kernels read quadrature.device_tables(self.nb_steps).  nb_steps <= 0 ("0 for random", the scripts then
        # call set_steps_nb per batch) gets NaN placeholders of the reference's shape; n >= 1 is checked when integrating.
        if nb_steps >= 1:
            w, s = compute_cc_weights(nb_steps)
            w, s = w.clone(), s.clone()
        else:
            w = torch.full((max(nb_steps + 1, 0), 1), float("nan"))
            s = w.clone()
        self.register_buffer("cc_weights", w)
        self.register_buffer("cc_steps", s)
        self.scaling = nn.Parameter(torch.zeros(input_size, device=self.device), requires_grad=False)

    def to(self, device):
        self.device = device
        super().to(device)
        return self

    # ------------------------------------------------------------------ core: one pass, both outputs
    def _transform(self, x, context=None, x0=None, want_jac=True, reverse_z=False, log_jac_in=None):
        """-> (z, log_jac or None).  One conditioner pass, one quadrature launch.  ``reverse_z`` / ``log_jac_in`` are
        the glue of a UMNNMAFFlow stack (UMNNMAFFlow.py:109-123): z with its dimensions reversed for the next block,
        log_jac added to the running sum -- inside the kernel on the HIP inference path, with ATen ops otherwise."""
        if self.solver not in _SOLVERS:
            return None, None
        integrand = self.net.parallel_nets
        h = self.net.make_embeding(x, context)
        d = x.shape[1]
        z0 = h.view(h.shape[0], -1, d)[:, 0, :]
        spec = mlp_spec(integrand)
        if self.nb_steps < 1:
            raise ValueError("UMNNMAF: nb_steps must be >= 1 when integrating (call set_steps_nb first)")
        # No-graph fast path only when nothing can ask for a gradient.  The reference's eval-mode direct integration
        # (UMNNMAF.py:89-105) stays differentiable through ordinary autograd; here eval + grad-requiring weights /
        # embedding goes through the same autograd Function as training.
        no_graph = (not torch.is_grad_enabled()) or not (
            x.requires_grad or h.requires_grad or (x0 is not None and x0.requires_grad)
            or any(p.requires_grad for p in integrand.parameters()))
        if _I._use_hip(spec, x):
            if no_graph and x0 is None:
                z, log_jac, _, _ = _I.hip_flow_block(spec, x, h, self.scaling, self.nb_steps, reverse_z, log_jac_in)
                return z, log_jac
            x0 = x0.to(x.device) if x0 is not None else None      # None = lower limit 0 inside the kernels
            if no_graph:
                F, fx, _ = _I.hip_forward(spec, x0, x, h, self.nb_steps)
            elif _I.fused_block_ok(x, h, self.scaling, x0, want_jac) and (log_jac_in is None or log_jac_in.dtype == torch.float32):
                # training: the whole block -- quadrature + epilogue, and their backward -- as ONE autograd node
                return _I.FlowBlockTransform.apply(x.contiguous(), integrand, h.contiguous(), self.scaling, self.nb_steps,
                                                   reverse_z, log_jac_in, *integrand.parameters())
            else:
                F, fx = IntegralWithJacobianParams.apply(x0, x, integrand, h, self.nb_steps, *integrand.parameters())
        else:
            x0 = x0.to(x.device) if x0 is not None else torch.zeros_like(x)
            if no_graph:
                with torch.no_grad():
                    F = _I.aten_forward(integrand, x0, x, h, self.nb_steps)
            else:
                F = _SOLVERS[self.solver].apply(x0, x, integrand, _flatten(integrand.parameters()), h, self.nb_steps)
            fx = integrand(x, h) if want_jac else None
        z = torch.exp(self.scaling).unsqueeze(0) * (F + z0)
        log_jac = torch.log(fx + 1e-10) + self.scaling.unsqueeze(0) if want_jac else None
        if reverse_z:
            z = torch.flip(z, [1])
        if log_jac_in is not None and log_jac is not None:
            log_jac = log_jac_in + log_jac
        return z, log_jac

    # ------------------------------------------------------------------ reference API
    def forward(self, x, method=None, x0=None, context=None):
        return self._transform(x, context, x0, want_jac=False)[0]

    def compute_log_jac(self, x, context=None):
        h = self.net.make_embeding(x, context)
        integrand = self.net.parallel_nets
        spec = mlp_spec(integrand)
        if _I._use_hip(spec, x):
            # f(x;h) is quadrature node 0: a one-step launch of the forward kernel evaluates it (two nodes) without the
            # [B*d, 1+E] row matrix the reference materialises (UMNNMAF.py:136-139, 263-284)
            wants_graph = torch.is_grad_enabled() and (x.requires_grad or h.requires_grad
                                                       or any(p.requires_grad for p in integrand.parameters()))
            if wants_graph:
                jac = IntegralWithJacobianParams.apply(None, x, integrand, h, 1, *integrand.parameters())[1]
            else:
                jac = _I.hip_forward(spec, None, x, h, 1)[1]
        else:
            jac = integrand(x, h)
        return torch.log(jac + 1e-10) + self.scaling.unsqueeze(0).expand(x.shape[0], -1)

    def compute_log_jac_bis(self, x, context=None):
        return self._transform(x, context)

    def compute_ll(self, x, context=None):
        z, log_jac = self._transform(x, context)
        z.clamp_(-10., 10.)
        log_prob_gauss = -.5 * (torch.log(self.pi * 2) + z ** 2).sum(1)
        return log_prob_gauss + log_jac.sum(1), z

    def compute_ll_bis(self, x, context=None):
        z, log_jac = self._transform(x, context)
        z.clamp_(-10., 10.)
        return log_jac, z

    def compute_bpp(self, x, alpha=1e-6, context=None):
        d = x.shape[1]
        ll, z = self.compute_ll(x, context=context)
        bpp = -ll / (d * np.log(2)) - np.log2(1 - 2 * alpha) + 8 \
            + 1 / d * (torch.log2(torch.sigmoid(x)) + torch.log2(1 - torch.sigmoid(x))).sum(1)
        z.clamp_(-10., 10.)
        return bpp, ll, z

    def set_steps_nb(self, nb_steps):
        # The registered cc_weights / cc_steps buffers stay at constructor shape (checkpoint round trip, see __init__);
        # every integral reads the tables of the CURRENT nb_steps from the per-device cache, so the reference's
        # stale-buffer crash in eval + CCParallel (UMNNMAF.py:48-50,104-106,172-173) cannot happen here.
        self.nb_steps = nb_steps

    def compute_lipschitz(self, nb_iter=10):
        return self.net.parallel_nets.compute_lipschitz(nb_iter)

    def force_lipschitz(self, L=1.5):
        self.net.parallel_nets.force_lipschitz(L)

    computeLL = compute_ll
    computell = compute_ll
    computeLipshitz = compute_lipschitz
    forceLipshitz = force_lipschitz
    forcei_lpschitz = force_lipschitz

    # ------------------------------------------------------------------ inversion (sampling), row f3
    def _conditional_integral(self, cand, h_j):
        """int_0^cand f(t; h_j) dt for one dimension: cand [R,1], h_j [R,E] -> [R,1]."""
        integrand = self.net.parallel_nets
        spec = mlp_spec(integrand)
        if _I._use_hip(spec, cand):
            return _I.hip_forward(spec, None, cand, h_j, self.nb_steps)[0]
        with torch.no_grad():
            return _I.aten_forward(lambda t, hh: integrand.independant_forward(torch.cat((t, hh), 1)),
                                   torch.zeros_like(cand), cand, h_j, self.nb_steps)

    def invert(self, z, iter=10, context=None):
        """Dimension-by-dimension bracket search: 10 candidates per round on [left,right] (starting at +-50), keep
        the sub-interval next to the candidate whose image is closest to the target (UMNNMAF.py:182-232)."""
        K = 10
        B, d = z.shape
        dev = z.device
        spec = mlp_spec(self.net.parallel_nets)
        if (_I._use_hip(spec, z) and z.dtype == torch.float32 and self.nb_steps >= 1 and iter >= 1 and B > 0
                and self.solver in _SOLVERS):
            # d x (conditioner + ONE launch): the whole 10-way / `iter`-round search of a dimension runs inside the kernel
            with torch.no_grad():
                z = z.contiguous()
                x_inv = torch.zeros(B, d, device=dev)
                scaling = self.scaling.detach().float().contiguous()
                done = True
                # Dimension j reads E of the E*d embedding entries.  Wide unconditional conditioners compute only those columns of
                # their last layer (MADE.raw_rows: for d = 784 that layer is 23 520 rows of which 30 are read) into a standing buffer
                made = self.net.made
                E = made.nout // made.nin
                restrict = (context is None and not isinstance(made, ConditionnalMADE) and self.net.embedding_dtype in (None, torch.float32)
                            and os.environ.get("UMNN_INVERT_ROWS", "1") != "0")
                rows_all = (torch.arange(E, device=dev) * d).view(1, E) + torch.arange(d, device=dev).view(d, 1) if restrict else None
                h_buf = None
                for j in range(self.input_size):
                    hj = made.raw_rows(x_inv, rows_all[j]) if restrict else None
                    if hj is not None:
                        if h_buf is None:
                            h_buf = torch.zeros(B, E * d, device=dev)
                        h_buf.view(B, E, d)[:, :, j] = hj
                        h = h_buf
                    else:
                        restrict = False
                        # umnn_flow_invert_dim reads an fp32 embedding (it has no umnn_io descriptor): a bf16 embedding
                        # (set_embedding_dtype, autocast) is widened here -- exact -- instead of being misread as fp32
                        h = self.net.make_embeding(x_inv, context).float().contiguous()
                    if not _I.hip_invert_dim(spec, h, z, scaling, self.nb_steps, j, iter, x_inv):
                        done = False
                        break
            if done:
                return x_inv
        frac = torch.linspace(0., 1., K, device=dev).view(K, 1)
        x_inv = torch.zeros(B, d, device=dev)
        scale = torch.exp(self.scaling)
        rows = torch.arange(B, device=dev)
        with torch.no_grad():
            for j in range(self.input_size):
                h = self.net.make_embeding(x_inv, context)
                h3 = h.view(B, -1, d)
                h_j = h3[:, :, j]                                   # [B,E]; row 0 doubles as the offset
                offset = h_j[:, 0]
                h_rep = h_j.unsqueeze(0).expand(K, -1, -1).reshape(K * B, -1)
                left = torch.full((B,), -50., device=dev)
                right = torch.full((B,), 50., device=dev)
                best = torch.zeros(B, device=dev)
                for _ in range(iter):
                    cand = frac * (right - left).unsqueeze(0) + left.unsqueeze(0)          # [K,B]
                    F = self._conditional_integral(cand.reshape(-1, 1), h_rep).view(K, B)
                    z_est = scale[j] * (offset.unsqueeze(0) + F)
                    m = torch.abs(z_est - z[:, j].unsqueeze(0)).argmin(0)                  # [B]
                    below = z_est[m, rows] < z[:, j]
                    lo = cand[(m - 1).clamp(min=0), rows]
                    hi = cand[(m + 1).clamp(max=K - 1), rows]
                    best = cand[m, rows]
                    left = torch.where(below, best, lo)
                    right = torch.where(below, hi, best)
                x_inv[:, j] = best
        return x_inv


class ListModule(object):
    """Registers modules on ``module`` as attributes ``prefix0, prefix1, ...`` and indexes them like a list."""

    def __init__(self, module, prefix, *args):
        self.module, self.prefix, self.num_module = module, prefix, 0
        for m in args:
            self.append(m)

    def append(self, new_module):
        if not isinstance(new_module, nn.Module):
            raise ValueError('Not a Module')
        self.module.add_module(self.prefix + str(self.num_module), new_module)
        self.num_module += 1

    def __len__(self):
        return self.num_module

    def __getitem__(self, i):
        if not 0 <= i < self.num_module:
            raise IndexError('Out of bound')
        return getattr(self.module, self.prefix + str(i))


class UMNNMAFFlow(nn.Module):
    def __init__(self, nb_flow=1, nb_in=1, hidden_derivative=[50, 50, 50, 50], hidden_embedding=[50, 50, 50, 50],
                 embedding_s=20, nb_steps=50, act_func='ELU', solver="CC", cond_in=0, device="cpu"):
        super().__init__()
        self.device = device
        self.register_buffer("pi", torch.tensor(math.pi))
        self.nets = ListModule(self, "Flow")
        for _ in range(nb_flow):
            emb = EmbeddingNetwork(nb_in, hidden_embedding, hidden_derivative, embedding_s, act_func=act_func,
                                   device=device, cond_in=cond_in).to(device)
            self.nets.append(UMNNMAF(emb, nb_in, nb_steps, device, solver=solver).to(device))

    def to(self, device):
        for net in self.nets:
            net.to(device)
        self.device = device
        super().to(device)
        return self

    def _stack(self, x, context, want_jac):
        """Run the blocks with the dimension reversal between them -> (z in original order, summed log_jac)."""
        log_jac = None
        nb = len(self.nets)
        for i, net in enumerate(self.nets):
            # every block but the last hands its z over reversed (the reference flips after every block and once more
            # at the end: the last two flips cancel); log_jac accumulates elementwise in each block's own input order
            x, lj = net._transform(x, context, want_jac=want_jac, reverse_z=i + 1 < nb, log_jac_in=log_jac)
            if want_jac:
                log_jac = lj
        return x, (log_jac if want_jac else 0.)

    def forward(self, x, context=None):
        return self._stack(x, context, False)[0]

    def invert(self, z, iter=10, context=None):
        z = torch.flip(z, [1])
        for i in range(len(self.nets) - 1, -1, -1):
            z = self.nets[i].invert(torch.flip(z, [1]), iter, context=context)
        return z

    def compute_log_jac(self, x, context=None):
        return self._stack(x, context, True)[1]

    def compute_log_jac_bis(self, x, context=None):
        return self._stack(x, context, True)

    def _one_pass_ok(self, x):
        """The fused one-pass log-likelihood (umnn_flow_ll_block_forward) applies: HIP path, fp32, nothing can ask for
        a gradient (same rule as UMNNMAF._transform)."""
        if len(self.nets) == 0 or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
            return False
        for net in self.nets:
            if net.solver not in _SOLVERS or net.nb_steps < 1 or not _I._use_hip(mlp_spec(net.net.parallel_nets), x):
                return False
            if net.net.embedding_dtype not in (None, torch.float32) or torch.is_autocast_enabled():
                return False                        # (the one-pass entry point is fp32-only; bf16 storage takes the _io route)
        if not torch.is_grad_enabled():
            return True
        return not (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def compute_ll(self, x, context=None):
        if self._one_pass_ok(x):
            # nb_flow x (conditioner + ONE launch): z, the running per-sample log-likelihood and the Gaussian term all
            # leave the quadrature kernel; no [B,d] log_jac accumulation, no elementwise epilogue (UMNNMAFFlow.py:109-119)
            with torch.no_grad():
                x = x.contiguous()
                ll = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
                scratch = torch.empty_like(x)
                cnt = _I.ll_counters(x.shape[0], x.device)      # (under a hipGraph capture: this call's own zeroed buffer)
                nb = len(self.nets)
                for i, net in enumerate(self.nets):
                    h = net.net.make_embeding(x, context)
                    x = _I.hip_flow_ll_block(mlp_spec(net.net.parallel_nets), x, h.contiguous(), net.scaling, net.nb_steps,
                                             reverse_z=i + 1 < nb, first=i == 0, last=i + 1 == nb, ll=ll, scratch=scratch, cnt=cnt)
            return ll, x
        z, log_jac = self._stack(x, context, True)
        if (z.is_cuda and z.dtype == torch.float32 and log_jac.dtype == torch.float32 and z.dim() == 2 and torch.is_grad_enabled()
                and (z.requires_grad or log_jac.requires_grad) and not torch.is_autocast_enabled()
                and os.environ.get("UMNN_FUSED_TRAIN", "1") != "0"):
            return _I.FlowLogLikelihood.apply(z, log_jac), z        # (training: the reduction and its backward as one launch each)
        log_prob_gauss = -.5 * (torch.log(self.pi * 2) + z ** 2).sum(1)
        return log_jac.sum(1) + log_prob_gauss, z

    def compute_ll_bis(self, x, context=None):
        z, log_jac = self._stack(x, context, True)
        return log_jac + -.5 * (torch.log(self.pi * 2) + z ** 2), z

    def compute_bpp(self, x, alpha=1e-6, context=None):
        d = x.shape[1]
        ll, z = self.compute_ll(x, context=context)
        bpp = -ll / (d * np.log(2)) - np.log2(1 - 2 * alpha) + 8 \
            + 1 / d * (torch.log2(torch.sigmoid(x)) + torch.log2(1 - torch.sigmoid(x))).sum(1)
        return bpp, ll, z

    def set_steps_nb(self, nb_steps):
        for net in self.nets:
            net.set_steps_nb(nb_steps)

    def set_embedding_dtype(self, dtype):
        """Storage type of the [B, E*d] embedding h between the conditioner and the quadrature kernels (an extension of
        this package, configuration C4): ``torch.bfloat16`` makes the conditioner's last GEMM write bf16 and the kernels
        read it with bf16 loads (fp32 arithmetic inside); ``None`` restores the conditioner's own dtype."""
        for net in self.nets:
            net.net.embedding_dtype = dtype

    def compute_lipschitz(self, nb_iter=10):
        L = 1.
        for net in self.nets:
            L *= net.compute_lipschitz(nb_iter)
        return L

    def force_lipschitz(self, L=1.5):
        for net in self.nets:
            net.force_lipschitz(L)

    computell = compute_ll
    computeLL = compute_ll
    computeLipshitz = compute_lipschitz
    forceLipshitz = force_lipschitz
    forcei_lpschitz = force_lipschitz
