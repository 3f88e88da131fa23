"""MADE conditioner (adjacent row a13 of the scope table): produces the embedding h the quadrature consumes.

Mirrors models/UMNN/made.py (MaskedLinear :16-27, MADE :30-144, ConditionnalMADE :146-195): constructor
signatures, ``net.{0,2,..}.{weight,bias,mask}`` state_dict keys and mask construction are the reference's.  The
GEMMs stay on PyTorch-ROCm (hipBLASLt): at the BSDS300 shape they are ~2 % of a block's FLOPs.
MI355X-side changes that do not alter results: the masked weight ``mask*W`` is cached between calls while the
weight is unchanged and no graph is being recorded, and ConditionnalMADE only computes the output columns it keeps.

Inference fast path (CUDA tensors, autograd off, env ``UMNN_MADE_BF16X3`` != 0): every masked linear runs as ONE bf16
GEMM with fp32 accumulation, ``[xh | xl | xh | 1 | 1] @ [Wh | Wh | Wl | bh | bl]^T`` (x = xh + xl, W = Wh + Wl in bf16
pieces): the fp32 hipBLASLt GEMM of the BSDS300 output layer takes 168 us, this one 60 us, max error 3e-6 of the output
range.  The left operand is built by the HIP kernel ``umnn_made_split3`` (csrc/made_split.hip) straight from the
previous layer's raw output (ReLU fused); the packed weights are cached like the masked weight.  Training (autograd on)
keeps the fp32 ``F.linear`` chain.
"""
import ctypes
import math
import os
import threading
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


_capture_state = threading.local()      # .cache_ok: a capture on THIS thread that tracks weight versions itself (GraphedLL)


class capture_may_cache:
    """Context manager (graphs.GraphedLL): the hipGraph being captured on this thread re-captures when a weight version
    moves, so the conditioner's cached masked / packed weights may be baked into it."""

    def __enter__(self):
        self._old = getattr(_capture_state, "cache_ok", False)
        _capture_state.cache_ok = True

    def __exit__(self, *exc):
        _capture_state.cache_ok = self._old


class MaskedLinear(nn.Linear):
    """nn.Linear whose weight is multiplied elementwise by a fixed 0/1 ``mask`` buffer."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self.register_buffer('mask', torch.ones(out_features, in_features))
        self._cache = None      # (weight version, weight data_ptr, masked weight)
        self._packed = None     # (key, bf16 [Wh|Wh|Wl|bh|bl] operand)
        self._frags = None      # (key, (bf16 MFMA fragments, fp32 bias)) for the fused conditioner kernel

    def set_mask(self, mask):
        # mask arrives [in, out] (numpy, bool); stored [out, in] like the weight
        self.mask.data.copy_(torch.from_numpy(np.ascontiguousarray(mask.T).astype(np.float32)))
        self._cache = None
        self._packed = None
        self._frags = None

    def invalidate_caches(self):
        """Drop the cached masked / packed weights.  The caches are keyed on tensor versions; writes through ``.data``
        (or any other route that does not bump ``_version``) are invisible to that key, so code doing such writes must
        call this (``umnn_amd.invalidate_caches(model)`` does it for a whole model)."""
        self._cache = None
        self._packed = None
        self._frags = None

    @staticmethod
    def _capturing(t):
        # a capture that tracks the weights' versions itself (graphs.GraphedLL re-captures when they move) may bake the caches in
        return t.is_cuda and not getattr(_capture_state, "cache_ok", False) and torch.cuda.is_current_stream_capturing()

    @staticmethod
    def _may_store(t):
        """Nothing produced while a stream capture is recording may enter a cache: its kernels have only been recorded, the tensor
        holds nothing until that graph is replayed, and an eager call (or a second capture) that found it under a matching key
        would read uninitialised weights.  (A capture that may bake the caches -- graphs.GraphedLL -- fills them with an eager
        run BEFORE it records; a miss inside the capture is then recomputed inside the graph, which is merely slower.)"""
        return not (t.is_cuda and torch.cuda.is_current_stream_capturing())

    def masked_weight(self):
        if torch.is_grad_enabled() and self.weight.requires_grad:
            return self.mask * self.weight
        if self._capturing(self.weight):
            # inside a hipGraph capture the product must be part of the graph: a cached tensor would be baked in as
            # a constant and replays after an optimizer step would read stale conditioner weights
            return (self.mask * self.weight).detach()
        key = (self.weight._version, self.weight.data_ptr(), self.mask._version, self.mask.data_ptr())
        if self._cache is None or self._cache[0] != key:
            w = (self.mask * self.weight).detach()
            if not self._may_store(self.weight):
                return w
            self._cache = (key, w)
        return self._cache[1]

    def packed_bf16(self, rows=None):
        """[Wh | Wh | Wl | bh | bl | 0-pad] as bf16 [N, pad8(3K+2)] for the K-concatenated bf16 GEMM (cached while
        weight, mask and bias are unchanged).  ``rows``: optional output-row selection (ConditionnalMADE)."""
        key = (self.weight._version, self.weight.data_ptr(), self.mask._version, self.mask.data_ptr(),
               self.bias._version, self.bias.data_ptr(), None if rows is None else rows.data_ptr())
        capturing = self._capturing(self.weight)        # see masked_weight: recompute inside the graph, never cache
        if capturing or self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                W, b = self.mask * self.weight, self.bias
                if rows is not None:
                    W, b = W.index_select(0, rows), b.index_select(0, rows)
                Wh, bh = W.bfloat16(), b.bfloat16()
                Wl, bl = (W - Wh.float()).bfloat16(), (b - bh.float()).bfloat16()
                K = W.shape[1]
                pad = (-(3 * K + 2)) % 8
                packed = torch.cat([Wh, Wh, Wl, bh[:, None], bl[:, None],
                                    torch.zeros(W.shape[0], pad, dtype=torch.bfloat16, device=W.device)], 1).contiguous()
            if capturing or not self._may_store(self.weight):
                return packed
            self._packed = (key, packed)
        return self._packed[1]

    def packed_fragments(self, rows=None):
        """(fragments, bias) for ``umnn_made_mlp_forward``: the masked weight as bf16 MFMA fragments
        [tile][K-step][piece hi/lo][lane][8] in the K order of include/umnn_cc.h, and the fp32 bias (cached while weight, mask
        and bias are unchanged).  ``rows``: optional output-row selection (ConditionnalMADE)."""
        key = (self.weight._version, self.weight.data_ptr(), self.mask._version, self.mask.data_ptr(),
               self.bias._version, self.bias.data_ptr(), None if rows is None else rows.data_ptr())
        capturing = self._capturing(self.weight)        # see masked_weight: recompute inside the graph, never cache
        if capturing or self._frags is None or self._frags[0] != key:
            with torch.no_grad():
                W, b = self.mask * self.weight, self.bias
                if rows is not None:
                    W, b = W.index_select(0, rows), b.index_select(0, rows)
                packed = (pack_fragments(W.float()), b.float().contiguous())
            if capturing or not self._may_store(self.weight):
                return packed
            self._frags = (key, packed)
        return self._frags[1]

    def forward(self, input):
        return F.linear(input, self.masked_weight(), self.bias)


_FRAG_INDEX = {}      # (N, K, device) -> gather indices of pack_fragments


def pack_fragments(W):
    """[N, K] fp32 -> bf16 [T, S, 2, 64, 8]: A-operand fragments of v_mfma_f32_16x16x32_bf16 for D[out][row] = W[out][k] act[k][row].
    Lane (g = lane >> 4, rho = lane & 15) of fragment (tile t, K-step s) holds, in slot j, W[16t + rho][32s + 16(j>>2) + 4g + (j&3)]
    (zero outside the matrix) -- the K order in which the fused kernel's accumulators are the next layer's operand; piece 0 is
    the value rounded to bf16, piece 1 the rounded remainder."""
    N, K = W.shape
    T, S = (N + 15) // 16, (K + 31) // 32
    key = (N, K, W.device)
    idx = _FRAG_INDEX.get(key)
    if idx is None:
        lane = torch.arange(64, device=W.device)
        g, rho = lane >> 4, lane & 15
        j = torch.arange(8, device=W.device)
        n = 16 * torch.arange(T, device=W.device).view(T, 1, 1, 1) + rho.view(1, 1, 64, 1)                       # [T,1,64,1]
        k = 32 * torch.arange(S, device=W.device).view(1, S, 1, 1) + (16 * (j >> 2) + (j & 3)).view(1, 1, 1, 8) \
            + 4 * g.view(1, 1, 64, 1)                                                                            # [1,S,64,8]
        idx = (n.expand(T, S, 64, 8) * (32 * S) + k.expand(T, S, 64, 8)).reshape(-1)
        # (never cached from inside a stream capture: the kernels that fill it have only been RECORDED then -- the tensor holds
        #  nothing until that graph is replayed, and an eager call that found it in the cache would gather with garbage)
        if not (W.is_cuda and torch.cuda.is_current_stream_capturing()):
            if len(_FRAG_INDEX) > 64:
                _FRAG_INDEX.clear()
            _FRAG_INDEX[key] = idx
    Wp = torch.zeros(16 * T, 32 * S, dtype=torch.float32, device=W.device)
    Wp[:N, :K] = W
    frag = Wp.reshape(-1).index_select(0, idx).view(T, S, 64, 8)
    hi = frag.bfloat16()
    lo = (frag - hi.float()).bfloat16()
    return torch.stack((hi, lo), 2).contiguous()


def invalidate_caches(module):
    """Drop every MaskedLinear cache under ``module`` (after writes that bypass tensor versioning, e.g. ``p.data``)."""
    for m in module.modules():
        if isinstance(m, MaskedLinear):
            m.invalidate_caches()


_FAST = {"ok": None, "enabled": os.environ.get("UMNN_MADE_BF16X3", "1") != "0"}


def set_made_fast_path(enabled):
    """Inference conditioner GEMMs as K-concatenated bf16 GEMMs (True, default) or plain fp32 ``F.linear`` (False)."""
    _FAST["enabled"] = bool(enabled)


def get_made_fast_path():
    return _FAST["enabled"]


def _fast_path_ok(x):
    """bf16x3 GEMM path: CUDA fp32 input, autograd off, HIP library present, torch.mm(out_dtype=) available."""
    if torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32:
        return False
    if not _FAST["enabled"]:
        return False
    if _FAST["ok"] is None:
        from . import _lib
        _lib.lib()              # a missing HIP library is an error on a GPU box (HipLibraryMissing), never a silent fallback
        try:
            a = torch.zeros(8, 8, dtype=torch.bfloat16, device=x.device)
            torch.mm(a, a.t(), out_dtype=torch.float32)
            _FAST["ok"] = True
        except (TypeError, RuntimeError, NotImplementedError) as e:     # a torch build without mm(out_dtype=) on this device
            _FAST["ok"] = False
            warnings.warn("umnn_amd: torch.mm(bf16, bf16, out_dtype=float32) is not available here "
                          f"({type(e).__name__}: {e}); the inference conditioner runs the fp32 F.linear chain instead of the "
                          "K-concatenated bf16 GEMMs.", RuntimeWarning, stacklevel=3)
    return _FAST["ok"]


def _fast_chain(a, layers, last_rows=None, out_dtype=None):
    """a [B, K0] fp32 -> output of the MaskedLinear/ReLU chain, every GEMM as one K-concatenated bf16 GEMM (fp32
    accumulation).  ``out_dtype=torch.bfloat16``: the LAST GEMM writes the embedding straight in bf16 (configuration C4:
    the [B, E*d] tensor the quadrature kernels then read with bf16 loads -- half the bytes of that round trip)."""
    from . import _lib
    lib = _lib.lib()
    raw = a.contiguous()
    stream = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
    with torch.cuda.device(a.device):
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            packed = layer.packed_bf16(last_rows if last else None)
            op = torch.empty(raw.shape[0], packed.shape[1], dtype=torch.bfloat16, device=a.device)
            _lib.check(lib.umnn_made_split3(raw.data_ptr(), raw.shape[0], raw.shape[1], 1 if i > 0 else 0,
                                            op.data_ptr(), op.shape[1], stream), "made_split3")
            if last and out_dtype == torch.bfloat16:
                raw = torch.mm(op, packed.t())                       # bf16 out, fp32 accumulate inside the GEMM
            else:
                raw = torch.mm(op, packed.t(), out_dtype=torch.float32)
    return raw


_FUSED = {"enabled": os.environ.get("UMNN_MADE_FUSED", "1") != "0", "max_rows": int(os.environ.get("UMNN_MADE_FUSED_MAX_ROWS", "32768")),
          "wide_out": os.environ.get("UMNN_MADE_FUSED_WIDE_OUT", "0") == "1",
          "hybrid": os.environ.get("UMNN_MADE_FUSED_HYBRID", "0") == "1",
          "layered": os.environ.get("UMNN_MADE_LAYERED", "1") != "0"}


def set_made_fused(enabled, max_rows=None, wide_out=None, hybrid=None, layered=None):
    """The whole conditioner of a block as ONE launch (``umnn_made_mlp_forward``) on the inference path when every width is
    <= 512 (see ``_fused_ok``) and the batch has at most ``max_rows`` rows (default 32768).  ``False``: always the per-layer
    path.  ``wide_out=True`` lifts the limit on the OUTPUT width (tests / measurements)."""
    _FUSED["enabled"] = bool(enabled)
    if max_rows is not None:
        _FUSED["max_rows"] = int(max_rows)
    if wide_out is not None:
        _FUSED["wide_out"] = bool(wide_out)
    if hybrid is not None:           # wide outputs: hidden stack in the kernel, output layer as one library GEMM (measured: no gain)
        _FUSED["hybrid"] = bool(hybrid)
    if layered is not None:          # wide outputs: one launch of the kernel per masked linear, grid over rows x output tiles
        _FUSED["layered"] = bool(layered)


def _fused_ok(a, layers, n_out=None):
    """-> 0 (split + library GEMM per layer), 1 (whole conditioner in one launch), 2 (hidden stack in one launch + the output
    layer as ONE library GEMM) or 3 (one launch of ``made_linear_kernel`` per masked linear).  Every input / hidden width must
    be <= 512.  Measured with bench.py's workloads (tools/bench_fused.sh, tools/made_layered_check.py): narrow outputs (toy 20,
    POWER 180 columns) win with the whole MLP in the kernel (0.203 -> 0.162 ms, 2.21 -> 2.00 ms per step); wide output layers
    (BSDS300's 1890, the VAE flow's 1920 columns) lose there (several passes of a kernel that streams its weights at one wave
    per SIMD: 12.92 vs 13.11 ms, 1.02 vs 1.34 ms), and feeding hipBLASLt's output GEMM from the kernel's hidden stack (mode 2,
    ``hybrid=True``) does not beat four short launches either (VAE 1.03 vs 1.11 ms, BSDS300 13.55 vs 13.58 ms); one launch per
    layer with the grid split over output tiles (mode 3) wins at launch-bound batch sizes (VAE at 1024 rows 1.00 -> 0.91 ms) and
    the hidden layers also at 8192 rows (C3: 140 -> 100 us per block with the 1890-column output layer back on split + hipBLASLt
    above ``UMNN_MADE_LAYERED_LIB_OUT_ROWS`` = 4096 rows); taken up to ``UMNN_MADE_LAYERED_MAX_ROWS`` rows (32768)."""
    if not _FUSED["enabled"] or a.shape[0] > _FUSED["max_rows"] or len(layers) > 8:
        return 0
    if not all(l.in_features <= 512 for l in layers):
        return 0
    n_out = layers[-1].out_features if n_out is None else n_out
    if n_out <= 512 or _FUSED.get("wide_out", False):
        return 1
    if len(layers) >= 2 and _FUSED.get("hybrid", False):
        return 2
    return 3 if (_FUSED.get("layered", True) and a.shape[0] <= _LAYERED_MAX_ROWS) else 0


def _fused_chain(a, layers, last_rows=None, out_dtype=None, mode=1):
    """a [B, K0] fp32 -> conditioner output through the fused kernel (csrc/made_fused.hip): the whole MLP (mode 1) or its
    hidden stack, whose last ReLU output leaves as the bf16 operand of the output layer's library GEMM (mode 2)."""
    from . import _lib
    lib = _lib.lib()
    a = a.contiguous()
    inner = layers if mode == 1 else layers[:-1]
    net = _lib.MadeNet()
    net.n_layers = len(inner)
    net.widths[0] = inner[0].in_features
    keep = []
    for i, layer in enumerate(inner):
        frags, bias = layer.packed_fragments(last_rows if (mode == 1 and i == len(inner) - 1) else None)
        keep += [frags, bias]
        net.widths[i + 1] = bias.shape[0]
        net.W[i], net.b[i] = frags.data_ptr(), bias.data_ptr()
    bf16 = out_dtype == torch.bfloat16
    with torch.cuda.device(a.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        if mode == 1:
            out = torch.empty(a.shape[0], net.widths[len(inner)], device=a.device, dtype=torch.bfloat16 if bf16 else torch.float32)
            _lib.check(lib.umnn_made_mlp_forward_ex(ctypes.byref(net), a.data_ptr(), a.shape[0], out.data_ptr(), 1 if bf16 else 0, 0,
                                                    stream), "umnn_made_mlp_forward")
            return out
        packed = layers[-1].packed_bf16(last_rows)                      # [N, pad8(3K+2)]: the per-layer path's weight operand
        op = torch.empty(a.shape[0], packed.shape[1], dtype=torch.bfloat16, device=a.device)
        _lib.check(lib.umnn_made_mlp_forward_ex(ctypes.byref(net), a.data_ptr(), a.shape[0], op.data_ptr(), 2, op.shape[1], stream),
                   "umnn_made_mlp_forward")
    if bf16:
        return torch.mm(op, packed.t())                                  # bf16 out, fp32 accumulate inside the GEMM
    return torch.mm(op, packed.t(), out_dtype=torch.float32)


_LAYERED_MAX_ROWS = int(os.environ.get("UMNN_MADE_LAYERED_MAX_ROWS", "32768"))
# above this many rows a wide OUTPUT layer goes back to split + library GEMM (hipBLASLt's large tiles win there: 67 + 9 us against
# 97 us at 8192 x 512 x 1890), the hidden layers stay on made_linear_kernel (11-13 us against 9 + 23 us each)
_LAYERED_LIB_OUT_ROWS = int(os.environ.get("UMNN_MADE_LAYERED_LIB_OUT_ROWS", "4096"))


def _layered_chain(a, layers, last_rows=None, out_dtype=None, a2=None):
    """a [B, K0] fp32 (or the two column blocks a | a2 of it) -> conditioner output, one launch of ``made_linear_kernel`` per
    masked linear (``umnn_made_linear_forward``: grid over row groups x output-tile groups, the bf16 split and the previous
    layer's ReLU in the operand load) -- the route of conditioners with a wide output layer at launch-bound batch sizes,
    replacing split + library GEMM pairs (two launches per layer) and the cat of ConditionnalMADE's two inputs."""
    from . import _lib
    lib = _lib.lib()
    cur, cur2 = a.contiguous(), (None if a2 is None else a2.contiguous())
    B = cur.shape[0]
    rt, fg = int(os.environ.get("UMNN_MADE_LINEAR_RT", "0")), int(os.environ.get("UMNN_MADE_LINEAR_G", "0"))
    with torch.cuda.device(a.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            bf16 = last and out_dtype == torch.bfloat16
            if last and i > 0 and B > _LAYERED_LIB_OUT_ROWS and layer.out_features > 512:
                packed = layer.packed_bf16(last_rows)                   # [N, pad8(3K+2)]: the library route's weight operand
                op = torch.empty(B, packed.shape[1], dtype=torch.bfloat16, device=a.device)
                _lib.check(lib.umnn_made_split3(cur.data_ptr(), B, cur.shape[1], 1, op.data_ptr(), op.shape[1], stream), "made_split3")
                return torch.mm(op, packed.t()) if bf16 else torch.mm(op, packed.t(), out_dtype=torch.float32)
            frags, bias = layer.packed_fragments(last_rows if last else None)
            out = torch.empty(B, bias.shape[0], device=a.device, dtype=torch.bfloat16 if bf16 else torch.float32)
            K = cur.shape[1] + (0 if cur2 is None else cur2.shape[1])
            _lib.check(lib.umnn_made_linear_forward(frags.data_ptr(), bias.data_ptr(), K, bias.shape[0], cur.data_ptr(),
                                                    None if cur2 is None else cur2.data_ptr(), cur.shape[1], B, 1 if i > 0 else 0,
                                                    out.data_ptr(), 1 if bf16 else 0, rt, fg, stream), "umnn_made_linear_forward")
            cur, cur2 = out, None
    return cur


# ---- training chain (round 6): the whole masked MLP as ONE autograd node ---------------------------------------------------------
# Under autograd the reference's chain (made.py:16-27,113-119) is, per masked linear, a mask product + F.linear + ReLU forward and a
# ReluBackward pass, two GEMMs, a column reduction for the bias gradient and the mask product's own backward: at the UCI shapes 15
# bias reductions of 16 us, 30 mask products and 20 ReLU passes per training step.  Here the chain is one node: the forward
# runs the same fp32 library GEMMs on the cached masked weight (one product per optimizer step, no autograd node), the ReLU in place;
# the backward runs, per layer, ONE pass that applies the ReLU mask to the incoming gradient and reduces the bias gradient in a fixed
# order (umnn_made_relu_bwd_bias: data-parallel replicas stay bit-identical), the two GEMMs, and the mask on the weight gradient in place.
_TRAIN_FUSED = {"enabled": os.environ.get("UMNN_MADE_TRAIN_FUSED", "1") != "0"}
_HAS_ADDMM_ACT = hasattr(torch, "_addmm_activation") and os.environ.get("UMNN_MADE_ADDMM_ACT", "1") != "0"


def _train_chain_ok(a, layers):
    if not (_TRAIN_FUSED["enabled"] and torch.is_grad_enabled() and a.is_cuda and a.dtype == torch.float32 and a.dim() == 2):
        return False
    if torch.is_autocast_enabled():
        return False
    if not all(l.weight.dtype == torch.float32 and l.bias is not None and l.weight.is_cuda for l in layers):
        return False
    return a.requires_grad or any(l.weight.requires_grad or l.bias.requires_grad for l in layers)


class _MadeTrainChain(torch.autograd.Function):
    """out = MaskedLinear_L(ReLU(... ReLU(MaskedLinear_1(a)))) [rows ``keep`` of the last layer only] as one node.
    apply(a, layers, keep, W_1, b_1, ..., W_L, b_L): ``layers`` the MaskedLinear modules (for their masks / cached masked weights),
    the parameters passed as tensors so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, a, layers, keep, *params):
        acts, weffs = [a.contiguous()], []
        cur = acts[0]
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            W, b = layer.masked_weight(), layer.bias.detach()        # (grad mode is off in here: the cached product, no autograd node)
            if last and keep is not None:
                W, b = W.index_select(0, keep), b.index_select(0, keep)
            weffs.append(W)
            if not last:
                # bias + ReLU in the GEMM's epilogue (hipBLASLt): no separate activation pass over [B, N]
                cur = y = torch._addmm_activation(b, cur, W.t(), use_gelu=False) if _HAS_ADDMM_ACT else torch.relu_(torch.addmm(b, cur, W.t()))
                acts.append(cur)
            else:
                y = torch.addmm(b, cur, W.t())
        ctx.layers, ctx.keep, ctx.n = layers, keep, len(layers)
        ctx.save_for_backward(*acts, *weffs)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib
        lib = _lib.lib()
        L = ctx.n
        acts, weffs = ctx.saved_tensors[:L], ctx.saved_tensors[L:]
        layers, keep = ctx.layers, ctx.keep
        g = g.contiguous()
        B = g.shape[0]
        grads = [None] * (2 * L)
        stream = ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
        with torch.cuda.device(g.device):
            for i in range(L - 1, -1, -1):
                last = i == L - 1
                layer = layers[i]
                N = g.shape[1]
                relu_out = None if last else acts[i + 1]
                need_b = ctx.needs_input_grad[3 + 2 * i + 1]
                if need_b or not last:
                    rb = int(lib.umnn_made_relu_bwd_bias_row_blocks(B, N))
                    partial = torch.empty(rb * N, device=g.device, dtype=torch.float32)
                    gb = torch.empty(N, device=g.device, dtype=torch.float32)
                    _lib.check(lib.umnn_made_relu_bwd_bias(g.data_ptr(), None if relu_out is None else relu_out.data_ptr(), B, N,
                                                           partial.data_ptr(), rb, gb.data_ptr(), stream), "umnn_made_relu_bwd_bias")
                    if need_b:
                        if last and keep is not None:
                            full = torch.zeros_like(layer.bias)
                            full.index_copy_(0, keep, gb)
                            gb = full
                        grads[2 * i + 1] = gb
                if ctx.needs_input_grad[3 + 2 * i]:
                    gW = torch.mm(g.t(), acts[i])
                    if last and keep is not None:
                        full = torch.zeros_like(layer.weight)
                        full.index_copy_(0, keep, gW.mul_(layer.mask.index_select(0, keep)))
                        gW = full
                    else:
                        gW.mul_(layer.mask)
                    grads[2 * i] = gW
                if i > 0 or ctx.needs_input_grad[0]:
                    g = torch.mm(g, weffs[i])
        return (g if ctx.needs_input_grad[0] else None, None, None, *grads)


def _train_chain(a, layers, keep=None):
    params = []
    for l in layers:
        params += [l.weight, l.bias]
    return _MadeTrainChain.apply(a, layers, keep, *params)


def _to_weight_dtype(x, layer):
    """bf16 / fp16 activations handed to fp32 weights outside autocast: widen (exact) instead of failing in F.linear."""
    if x.dtype != layer.weight.dtype and not torch.is_autocast_enabled():
        return x.to(layer.weight.dtype)
    return x


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

    def raw(self, x, out_dtype=None):
        """The masked MLP itself (what the flow's EmbeddingNetwork needs, whatever nout is)."""
        x = _to_weight_dtype(x, self.net[0])
        if _fast_path_ok(x):
            layers = [l for l in self.net if isinstance(l, MaskedLinear)]
            mode = _fused_ok(x, layers)
            if mode == 3:
                return _layered_chain(x, layers, out_dtype=out_dtype)
            if mode:
                return _fused_chain(x, layers, out_dtype=out_dtype, mode=mode)
            return _fast_chain(x, layers, out_dtype=out_dtype)
        layers = [l for l in self.net if isinstance(l, MaskedLinear)]
        out = _train_chain(x, layers) if _train_chain_ok(x, layers) else self.net(x)
        return out.to(out_dtype) if out_dtype is not None and out.dtype != out_dtype else out

    def raw_rows(self, x, rows):
        """Output COLUMNS ``rows`` (an int64 device tensor) of ``raw(x)`` only -> [B, len(rows)] fp32, or None when this
        conditioner / call is not one the restriction pays for (the caller then takes ``raw(x)``).  Sampling needs, per flow dimension
        j, the E embedding entries of that dimension out of E*d (UMNNMAF.invert, UMNNMAF.py:195-231, recomputes the whole MADE per
        dimension): for d = 784 the last masked linear is 23 520 rows of which 30 are read.  Inference path only (K-concatenated
        bf16 GEMMs): hidden layers as in ``raw``, the last layer as [B, 3K+2] x [3K+2, len(rows)] on the selected rows of its cached
        packed weight."""
        layers = [l for l in self.net if isinstance(l, MaskedLinear)]
        x = _to_weight_dtype(x, self.net[0])
        if len(layers) < 2 or layers[-1].out_features < 4096 or not _fast_path_ok(x):
            return None
        from . import _lib
        lib = _lib.lib()
        raw = x.contiguous()
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with torch.cuda.device(x.device):
            for i, layer in enumerate(layers):
                last = i == len(layers) - 1
                packed = layer.packed_bf16(None)
                if last:
                    packed = packed.index_select(0, rows)
                op = torch.empty(raw.shape[0], packed.shape[1], dtype=torch.bfloat16, device=x.device)
                _lib.check(lib.umnn_made_split3(raw.data_ptr(), raw.shape[0], raw.shape[1], 1 if i > 0 else 0,
                                                op.data_ptr(), op.shape[1], stream), "made_split3")
                raw = torch.mm(op, packed.t(), out_dtype=torch.float32)
        return raw

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
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                # (a pageable host -> device copy cannot be recorded: graphs._prime_for_capture builds the indices before it captures)
                raise RuntimeError("ConditionnalMADE: the kept-row indices must exist before a stream capture starts "
                                   "(umnn_amd.GraphedLL / GraphedTrainStep prime them; call model(x) once eagerly otherwise)")
            self._keep = idx.reshape(-1).to(device)
        return self._keep

    def raw(self, x, context, out_dtype=None):
        if context.dtype != x.dtype:
            context = context.to(x.dtype)
        if x.dtype == context.dtype == torch.float32 == self.net[0].weight.dtype and _fast_path_ok(x):
            layers = [l for l in self.net if isinstance(l, MaskedLinear)]
            if _fused_ok(x, layers, self.nin_non_cond * (self.nout // self.nin)) == 3:    # the kernel reads both blocks: no cat
                return _layered_chain(context, layers, self._kept_rows(x.device), out_dtype, a2=x)
        a = _to_weight_dtype(torch.cat((context, x), 1), self.net[0])
        if _fast_path_ok(a):
            layers = [l for l in self.net if isinstance(l, MaskedLinear)]
            mode = _fused_ok(a, layers, self.nin_non_cond * (self.nout // self.nin))
            if mode == 3:
                return _layered_chain(a, layers, self._kept_rows(a.device), out_dtype)
            if mode:
                return _fused_chain(a, layers, self._kept_rows(a.device), out_dtype, mode=mode)
            return _fast_chain(a, layers, self._kept_rows(a.device), out_dtype)
        keep = self._kept_rows(a.device)
        mlayers = [l for l in self.net if isinstance(l, MaskedLinear)]
        if _train_chain_ok(a, mlayers):
            out = _train_chain(a, mlayers, keep)
            return out.to(out_dtype) if out_dtype is not None and out.dtype != out_dtype else out
        layers = list(self.net)
        for layer in layers[:-1]:
            a = layer(a)
        last = layers[-1]
        out = F.linear(a, last.masked_weight().index_select(0, keep), last.bias.index_select(0, keep))
        return out.to(out_dtype) if out_dtype is not None and out.dtype != out_dtype else out

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
