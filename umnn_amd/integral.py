"""The quadrature operator: ``integrate``, ``ParallelNeuralIntegral`` and ``NeuralIntegral``.

Same names, argument order and gradient convention as the reference
(models/UMNN/ParallelNeuralIntegral.py:37-123, models/UMNN/NeuralIntegral.py:37-99):

    ParallelNeuralIntegral.apply(x0, x, integrand, flat_params, h, nb_steps=20, inv_f=False)
    NeuralIntegral.apply(x0, x, integrand, flat_params, h, nb_steps=20)
    integrate(x0, nb_steps, step_sizes, integrand, h, compute_grad=False, x_tot=None, inv_f=False, ...)

backward returns (-f(x0;h)*g, f(x;h)*g, None, d_theta, d_h, None[, None]) -- the Leibniz derivatives for the
limits and the VJP of the integrand over the nodes for theta and h, exactly the reference's formulas.

Two execution paths, chosen per call, never silently:
  * HIP: the integrand is an MLP ``nets.mlp_spec`` recognises and the tensors live on a GPU.  One fused gfx950
    kernel per direction through the C ABI (include/umnn_cc.h).  If libumnn_cc.so is missing this raises.
    Both solvers ("CC" sequential, "CCParallel" materialised) are the same arithmetic, so both classes land on the
    same kernels; the node axis is never materialised.
  * generic ATen: arbitrary callables (lambdas, custom modules -- reference tests/test_numerical_validation.py:33-41,
    UMNNMAF.invert :207) cannot be compiled; they are integrated with torch ops on whatever device they live on,
    in node chunks so memory stays bounded.  MLP integrands on host tensors also take this path (the kernels
    need device memory); ``path_taken()`` reports which path the last call used so tests can assert on it.
"""
import ctypes
import os
import weakref
import threading
import warnings

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from .nets import mlp_spec
from .quadrature import compute_cc_weights, device_tables

_state = threading.local()           # per-thread: last path taken, force_generic nesting
_warned = set()                       # fallbacks announced once per process (see _warn_once)


def _warn_once(key, message):
    """Every fallback off the HIP kernels is announced the first time it is taken (and ``path_taken()`` reports it)."""
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, RuntimeWarning, stacklevel=3)


_last_backward = {"path": None}      # process-wide: backward runs on autograd's worker threads, not on the caller's


def path_taken():
    """'hip' or 'aten': which path the calling thread's most recent forward (or directly called backward helper) used."""
    return getattr(_state, "path", None)


def backward_path_taken():
    """'hip' or 'aten': the path of the most recent quadrature BACKWARD in this process.  (autograd runs backward on its own
    worker threads, so the thread-local ``path_taken()`` of the caller does not see it.)"""
    return _last_backward["path"]


class force_generic:
    """Context manager: integrate MLP integrands with the generic ATen path too (A/B comparisons).  Thread-local: only the
    calling thread's integrals are rerouted (autograd's backward worker threads are told through the saved context)."""

    def __enter__(self):
        self._old = getattr(_state, "force_generic", False)
        _state.force_generic = True

    def __exit__(self, *exc):
        _state.force_generic = self._old


def _flatten(sequence):
    flat = [p.contiguous().view(-1) for p in sequence]
    return torch.cat(flat) if len(flat) > 0 else torch.tensor([])


# ----------------------------------------------------------------------------------------------
# HIP path
# ----------------------------------------------------------------------------------------------
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_desc_cache = {}      # id(spec.linears[0]) -> (key, MlpDesc, keep-alive tensors)


def _desc(spec):
    """struct umnn_mlp for the current weights.  The weights are read live (optimizer steps / force_lipschitz write in
    place and are seen through the same pointers); the ctypes struct itself is rebuilt only when a pointer moved."""
    lins = spec.linears
    key = tuple((lin.weight.data_ptr(), lin.bias.data_ptr()) for lin in lins) + (spec.hidden_act, spec.out_act)
    slot = id(lins[0])
    hit = _desc_cache.get(slot)
    if hit is not None and hit[0] == key and hit[3] is lins[0]:
        return hit[1], hit[2]
    d = _lib.MlpDesc()
    d.n_linear = len(lins)
    d.widths[0] = lins[0].in_features
    keep = []
    cacheable = True
    for l, lin in enumerate(lins):
        w, b = lin.weight.detach(), lin.bias.detach()
        if not w.is_contiguous():
            w, cacheable = w.contiguous(), False       # a copy: its contents would go stale
        if not b.is_contiguous():
            b, cacheable = b.contiguous(), False
        keep += [w, b]
        d.widths[l + 1] = lin.out_features
        d.W[l], d.b[l] = w.data_ptr(), b.data_ptr()
    d.hidden_act, d.out_act = spec.hidden_act, spec.out_act
    if cacheable:
        if len(_desc_cache) > 256:
            _desc_cache.clear()
        _desc_cache[slot] = (key, d, keep, lins[0])
    return d, keep


def _use_hip(spec, x):
    if spec is None or getattr(_state, "force_generic", False):
        return False
    if not x.is_cuda:
        _warn_once("host", "umnn_amd: MLP integrand on host tensors -> generic ATen quadrature "
                           "(the HIP kernels need GPU tensors; move the model and data to 'cuda').")
        return False
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return False
    if spec.linears[0].weight.device != x.device:
        raise RuntimeError("umnn_amd: integrand weights and inputs are on different devices")
    return True


_bwd_kind = {}
_BWD_WIDE = {"hip": os.environ.get("UMNN_BWD_WIDE", "") == "hip"}     # read once; set_backward_wide() at run time


def set_backward_wide(force_hip):
    """True: nets whose HIP backward only has the register-spilling generic wide kernels use them anyway (default: the
    materialised ATen chain on the GPU; environment UMNN_BWD_WIDE=hip at import)."""
    _BWD_WIDE["hip"] = bool(force_hip)


def _hip_backward_ok(spec, x, h):
    """False for nets the HIP backward only covers with its register-spilling generic wide variants: since round 3 only deep
    nets whose zero-padded weight images exceed the LDS (four or more hidden layers above 103 units, five above 63) --
    unequal widths up to 127 otherwise run the shape-exact fp32 kernels zero-padded (cc_backward.hip pad_to_exact_family),
    and a wide FIRST hidden layer over a narrow rest (MNISTExperiment's 100-50-50-50-50) has the three-stage kernels of
    cc_backward_front.hip (both backward precisions since round 4).  ``UMNN_BWD_WIDE=hip`` forces the HIP kernels anyway."""
    if _BWD_WIDE["hip"]:
        return True
    E = h.shape[1] // x.shape[1]
    key = (tuple((m.in_features, m.out_features) for m in spec.linears), E, _lib.get_option("bwd_precision"))
    kind = _bwd_kind.get(key)
    if kind is None:
        desc, keep = _desc(spec)
        kind = _lib.lib().umnn_cc_backward_kind(ctypes.byref(desc), E)
        if kind < -1:       # an error code (invalid descriptor, umnn_prepare_mlp failure): surfaced, never rerouted as "a wide net"
            _lib.check(kind, "umnn_cc_backward_kind")
        _bwd_kind[key] = kind
    if kind < 0:
        _warn_once(("bwd-aten", key[0]),
                   f"umnn_amd: the HIP backward has no shape-exact kernel for integrand widths {[w for _, w in key[0]]} "
                   "(deep net of unequal wide hidden layers: its zero-padded weight images exceed the LDS): differentiating with the materialised ATen chain on the "
                   "GPU instead (forward stays on the HIP kernel; umnn_amd.set_backward_wide(True) forces the HIP kernels).")
    return kind >= 0


def aten_backward_jac(integrand, x0, x, h, gF, gfx, nb_steps):
    """ATen counterpart of hip_backward for IntegralWithJacobian: the reference's quadrature VJP for the cotangent of F
    plus ordinary autograd through f(x;h) for the cotangent of f_x -> (dx0, dx, dh, dtheta_flat)."""
    dtheta, dh = aten_backward(integrand, x0, x, h, gF, nb_steps)
    dh = dh.view(h.shape)
    params = list(integrand.parameters())
    xr, hr = x.detach().requires_grad_(True), h.detach().requires_grad_(True)
    with torch.enable_grad():
        fx = integrand(xr, hr)
    dx = fx.detach() * gF
    if gfx is not None:
        grads = torch.autograd.grad(fx, [xr, hr] + params, gfx, allow_unused=True)
        dx = dx + grads[0]
        dh = dh + grads[1]
        dtheta = dtheta + _flatten([g if g is not None else torch.zeros_like(p) for g, p in zip(grads[2:], params)])
    with torch.no_grad():
        dx0 = -integrand(x0, h) * gF
    return dx0, dx, dh, dtheta


def _shape(spec, x, h):
    if x.dim() != 2 or h.dim() != 2 or h.shape[0] != x.shape[0]:
        raise RuntimeError("umnn_amd: expected x [B,d] and h [B,E*d]")
    B, d = x.shape
    E = h.shape[1] // d
    if E * d != h.shape[1] or spec.linears[0].in_features != 1 + E:
        raise RuntimeError(f"umnn_amd: h has {h.shape[1]} columns; the integrand expects (in_features-1)*d = "
                           f"{(spec.linears[0].in_features - 1) * d}")
    return B, d, E


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _io_prep(x_like, h):
    """Storage plan of one call (configuration C4: bf16 activations): tensors are handed to the kernels in the dtype the
    caller stores them in when that is fp32 or bf16 (no conversion pass, bf16 loads/stores inside the kernels); fp16 and
    anything else is converted to fp32 at the boundary.  -> (x dtype, h dtype, umnn_io or None)."""
    xd = x_like.dtype if x_like.dtype in (torch.float32, torch.bfloat16) else torch.float32
    hd = h.dtype if h.dtype in (torch.float32, torch.bfloat16) else torch.float32
    if xd == torch.float32 and hd == torch.float32:
        return xd, hd, None
    io = _lib.IoDesc(_lib.DTYPE_BF16 if xd == torch.bfloat16 else _lib.DTYPE_F32,
                     _lib.DTYPE_BF16 if hd == torch.bfloat16 else _lib.DTYPE_F32)
    return xd, hd, io


def _as(t, dtype):
    return None if t is None else t.detach().to(dtype).contiguous()


def hip_forward(spec, x0, x, h, nb_steps, inv_f=False):
    """-> (F, f_x, f_x0), each [B,d] in x's dtype.  x0 may be None (zeros)."""
    lib = _lib.lib()
    B, d, E = _shape(spec, x, h)
    out_dtype = x.dtype
    xd, hd, io = _io_prep(x, h)
    x, h, x0 = _as(x, xd), _as(h, hd), _as(x0, xd)
    w, s = device_tables(nb_steps, x.device)
    F, fx, fx0 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    desc, keep = _desc(spec)
    with torch.cuda.device(x.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if io is None:
            rc = lib.umnn_cc_forward(ctypes.byref(desc), _ptr(x0), _ptr(x), _ptr(h), _ptr(w), _ptr(s), int(nb_steps),
                                     B, d, E, int(bool(inv_f)), _ptr(F), _ptr(fx), _ptr(fx0), stream)
        else:
            rc = lib.umnn_cc_forward_io(ctypes.byref(desc), ctypes.byref(io), _ptr(x0), _ptr(x), _ptr(h), _ptr(w), _ptr(s),
                                        int(nb_steps), B, d, E, int(bool(inv_f)), _ptr(F), _ptr(fx), _ptr(fx0), stream)
    _lib.check(rc, "umnn_cc_forward")
    _state.path = "hip"
    if out_dtype != xd:                   # fp16 callers: fp32 inside, their dtype outside
        F, fx, fx0 = F.to(out_dtype), fx.to(out_dtype), fx0.to(out_dtype)
    return F, fx, fx0


# z_2 handed from the training forward to the backward (three-stage family): memory held from forward to backward.  Two caps: per
# block (UMNN_SAVE_Z2_MAX_GB, default 2) and over everything alive at once -- all blocks of a flow between its forward and its
# backward -- (UMNN_SAVE_Z2_TOTAL_GB, default 8; the 5-block MNISTExperiment flow at B = 100 holds 4.2 GB).  Above either, that block's
# backward recomputes z_2 (its stage A) as before: a speed / peak-memory trade, never a different result.
_Z2_MAX_BYTES = int(float(os.environ.get("UMNN_SAVE_Z2_MAX_GB", "2")) * (1 << 30))
_Z2_TOTAL_BYTES = int(float(os.environ.get("UMNN_SAVE_Z2_TOTAL_GB", "8")) * (1 << 30))
_z2_live = [0]


def _z2_release(nbytes):
    _z2_live[0] -= nbytes


def hip_flow_block(spec, x, h, scaling, nb_steps, reverse_z=False, log_jac_in=None, save_z2=False):
    """Fused block epilogue -> (z, log_jac, f_x, f_x0).  ``reverse_z``: z comes back with its dimensions reversed (the
    flip between the blocks of a flow); ``log_jac_in``: running log_jac of the previous blocks, added in the kernel.
    ``save_z2`` (training forward): -> (z, log_jac, f_x, f_x0, z2) where z2 is the buffer umnn_cc_backward_saved wants, or None when
    the pair does not apply to this net / arithmetic mode or the buffer would exceed UMNN_SAVE_Z2_MAX_GB (default 2) per block."""
    lib = _lib.lib()
    B, d, E = _shape(spec, x, h)
    if save_z2:
        z2 = None
        # (fp32 x-class tensors; the embedding h in fp32 or -- configuration C4 -- bf16, loaded as such by the kernels)
        if x.dtype == torch.float32 and h.dtype in (torch.float32, torch.bfloat16):
            desc, keep = _desc(spec)
            nfl = int(lib.umnn_cc_forward_z2_floats(ctypes.byref(desc), B, d, E, int(nb_steps)))
            if 0 < 4 * nfl <= _Z2_MAX_BYTES and _z2_live[0] + 4 * nfl <= _Z2_TOTAL_BYTES:
                _, _, io = _io_prep(x, h)
                x, h, scaling = _f32c(x), h.detach().contiguous(), _f32c(scaling)
                w, s = device_tables(nb_steps, x.device)
                z2 = torch.empty(nfl, device=x.device, dtype=torch.float32)
                _z2_live[0] += 4 * nfl
                weakref.finalize(z2, _z2_release, 4 * nfl)
                z, lj, fx, fx0 = (torch.empty_like(x) for _ in range(4))
                lj_in = _as(log_jac_in, torch.float32)
                with torch.cuda.device(x.device):
                    stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
                    if io is None:
                        rc = lib.umnn_flow_stack_block_forward_save(ctypes.byref(desc), _ptr(x), _ptr(h), _ptr(scaling), _ptr(w), _ptr(s),
                                                                    int(nb_steps), B, d, E, 1 if reverse_z else 0, _ptr(lj_in), _ptr(z), _ptr(lj),
                                                                    _ptr(fx), _ptr(fx0), _ptr(z2), nfl, stream)
                    else:
                        rc = lib.umnn_flow_stack_block_forward_save_io(ctypes.byref(desc), ctypes.byref(io), _ptr(x), _ptr(h), _ptr(scaling),
                                                                       _ptr(w), _ptr(s), int(nb_steps), B, d, E, 1 if reverse_z else 0,
                                                                       _ptr(lj_in), _ptr(z), _ptr(lj), _ptr(fx), _ptr(fx0), _ptr(z2), nfl, stream)
                if rc == 0:
                    _state.path = "hip"
                    return z, lj, fx, fx0, z2
                if rc != _lib.EUNSUPPORTED:
                    _lib.check(rc, "umnn_flow_stack_block_forward_save")
        return (*hip_flow_block(spec, x, h, scaling, nb_steps, reverse_z, log_jac_in), None)
    out_dtype = x.dtype
    xd, hd, io = _io_prep(x, h)
    x, h, scaling = _as(x, xd), _as(h, hd), _f32c(scaling)
    w, s = device_tables(nb_steps, x.device)
    z, lj, fx, fx0 = (torch.empty_like(x) for _ in range(4))
    desc, keep = _desc(spec)
    with torch.cuda.device(x.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        lj_in = _as(log_jac_in, xd)
        if io is None:
            rc = lib.umnn_flow_stack_block_forward(ctypes.byref(desc), _ptr(x), _ptr(h), _ptr(scaling), _ptr(w), _ptr(s),
                                                   int(nb_steps), B, d, E, 1 if reverse_z else 0, _ptr(lj_in),
                                                   _ptr(z), _ptr(lj), _ptr(fx), _ptr(fx0), stream)
        else:
            rc = lib.umnn_flow_stack_block_forward_io(ctypes.byref(desc), ctypes.byref(io), _ptr(x), _ptr(h), _ptr(scaling),
                                                      _ptr(w), _ptr(s), int(nb_steps), B, d, E, 1 if reverse_z else 0,
                                                      _ptr(lj_in), _ptr(z), _ptr(lj), _ptr(fx), _ptr(fx0), stream)
    _lib.check(rc, "umnn_flow_stack_block_forward")
    _state.path = "hip"
    if out_dtype != xd:
        z, lj, fx, fx0 = z.to(out_dtype), lj.to(out_dtype), fx.to(out_dtype), fx0.to(out_dtype)
    return z, lj, fx, fx0


def hip_invert_dim(spec, h, z, scaling, nb_steps, j, iters, x_inv):
    """Bracket search of flow dimension j for every sample in ONE launch (umnn_flow_invert_dim): writes x_inv[:, j].
    Returns False when the library has no kernel for this net (single hidden layer / LDS): the caller keeps its own loop."""
    lib = _lib.lib()
    B, d = z.shape
    E = h.shape[1] // d
    if E * d != h.shape[1] or spec.linears[0].in_features != 1 + E:
        raise RuntimeError("umnn_amd: embedding width does not match the integrand")
    w, s = device_tables(nb_steps, z.device)
    desc, keep = _desc(spec)
    with torch.cuda.device(z.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        rc = lib.umnn_flow_invert_dim(ctypes.byref(desc), _ptr(h), _ptr(z), _ptr(scaling), _ptr(w), _ptr(s), int(nb_steps),
                                      B, d, E, int(j), int(iters), _ptr(x_inv), stream)
    if rc == _lib.EUNSUPPORTED:
        _warn_once(("invert-host", tuple(l.out_features for l in spec.linears), _lib.get_forward_precision()),
                   "umnn_amd: no in-kernel inversion for this integrand / arithmetic mode "
                   f"({_lib.lib().umnn_last_error().decode('utf-8', 'replace')}): UMNNMAF.invert runs the host-driven bracket "
                   "search (d x iter forward launches per block).")
        return False
    _lib.check(rc, "umnn_flow_invert_dim")
    _state.path = "hip"
    return True


_row_counters = {}     # (device index, stream handle) -> zeroed uint32 [>= B] arrival counters (kernel leaves them zero)


def _counters(B, device, stream_handle):
    """Row-arrival counters of the one-pass log-likelihood.  Eager calls share one zeroed buffer per (device, stream): the
    finishing wave of every row resets its counter, so the buffer is all-zero again when the launch retires.  Under a
    hipGraph capture nothing may be created-and-cached (a tensor born in a capture lives in that graph's private pool and
    its zero-fill is a captured node, not an executed one): ``compute_ll`` asks ``fresh_counters`` for a buffer per captured
    call instead, whose memset is then part of every replay of that graph."""
    key = (device.index, stream_handle)
    t = _row_counters.get(key)
    if t is None or t.numel() < B:
        assert not torch.cuda.is_current_stream_capturing()
        t = _row_counters[key] = torch.zeros(max(B, 1024), dtype=torch.int32, device=device)
    return t


def ll_counters(B, device):
    """Counter buffer for one ``UMNNMAFFlow.compute_ll`` call on the current stream (see _counters)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(B, dtype=torch.int32, device=device)        # this graph's own buffer and memset node
    return _counters(B, device, torch.cuda.current_stream(device).cuda_stream)


def hip_flow_ll_block(spec, x, h, scaling, nb_steps, reverse_z, first, last, ll, scratch, cnt=None):
    """One link of UMNNMAFFlow.compute_ll with the whole log-likelihood arithmetic inside the launch
    (umnn_flow_ll_block_forward): -> z; ``ll`` [B] is updated in place, ``scratch`` [B,d] is this launch's own; ``cnt`` the
    zeroed int32 [>= B] row-arrival counters (``ll_counters``; the launch leaves them zero)."""
    lib = _lib.lib()
    B, d, E = _shape(spec, x, h)
    z = torch.empty_like(x)
    w, s = device_tables(nb_steps, x.device)
    desc, keep = _desc(spec)
    with torch.cuda.device(x.device):
        handle = torch.cuda.current_stream(x.device).cuda_stream
        if cnt is None:
            cnt = ll_counters(B, x.device)
        rc = lib.umnn_flow_ll_block_forward(ctypes.byref(desc), _ptr(x), _ptr(h), _ptr(scaling), _ptr(w), _ptr(s),
                                            int(nb_steps), B, d, E, 1 if reverse_z else 0, 1 if first else 0,
                                            1 if last else 0, _ptr(z), _ptr(scratch), _ptr(ll), _ptr(cnt),
                                            ctypes.c_void_p(handle))
    _lib.check(rc, "umnn_flow_ll_block_forward")
    _state.path = "hip"
    return z


def hip_backward(spec, x0, x, h, g, g_fx, nb_steps, need=(True, True, True, True), inv_f=False, z2_saved=None):
    """-> (dx0, dx, dh, dtheta_flat); entries are None where need[...] is False.  dx0/dx come back in x's dtype, dh in
    h's, dtheta in fp32 (the weights' dtype).  inv_f: the operator integrated 1/f (ParallelNeuralIntegral.py:58-59,70-72)."""
    lib = _lib.lib()
    B, d, E = _shape(spec, x, h)
    x_dtype, h_dtype = x.dtype, h.dtype
    xd, hd, io = _io_prep(x, h)
    x, h, g, x0, g_fx = _as(x, xd), _as(h, hd), _as(g, xd), _as(x0, xd), _as(g_fx, xd)
    w, s = device_tables(nb_steps, x.device)
    dx0 = torch.empty_like(x) if need[0] else None
    dx = torch.empty_like(x) if need[1] else None
    dh = torch.empty_like(h) if need[2] else None
    n_params = sum(l.weight.numel() + l.bias.numel() for l in spec.linears)
    dtheta = torch.empty(n_params, device=x.device, dtype=torch.float32) if need[3] else None
    desc, keep = _desc(spec)
    with torch.cuda.device(x.device):
        nbytes = lib.umnn_cc_backward_workspace_bytes(ctypes.byref(desc), B, d, E)
        ws = torch.empty(max(int(nbytes), 4), device=x.device, dtype=torch.uint8)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if z2_saved is not None and io is not None and not inv_f and x0 is None and not need[0]:
            rc = lib.umnn_cc_backward_saved_io(ctypes.byref(desc), ctypes.byref(io), _ptr(x), _ptr(h), _ptr(g), _ptr(g_fx), _ptr(w), _ptr(s),
                                               int(nb_steps), B, d, E, _ptr(dx), _ptr(dh), _ptr(dtheta), _ptr(z2_saved), int(z2_saved.numel()),
                                               _ptr(ws), int(nbytes), stream)
        elif z2_saved is not None and io is None and not inv_f and x0 is None and not need[0]:      # (that entry point has no d_x0 output)
            rc = lib.umnn_cc_backward_saved(ctypes.byref(desc), _ptr(x), _ptr(h), _ptr(g), _ptr(g_fx), _ptr(w), _ptr(s), int(nb_steps),
                                            B, d, E, _ptr(dx), _ptr(dh), _ptr(dtheta), _ptr(z2_saved), int(z2_saved.numel()),
                                            _ptr(ws), int(nbytes), stream)
        elif io is None and not inv_f:
            rc = lib.umnn_cc_backward(ctypes.byref(desc), _ptr(x0), _ptr(x), _ptr(h), _ptr(g), _ptr(g_fx),
                                      _ptr(w), _ptr(s), int(nb_steps), B, d, E,
                                      _ptr(dx0), _ptr(dx), _ptr(dh), _ptr(dtheta), _ptr(ws), int(nbytes), stream)
        else:
            rc = lib.umnn_cc_backward_io(ctypes.byref(desc), ctypes.byref(io) if io is not None else None,
                                         _ptr(x0), _ptr(x), _ptr(h), _ptr(g), _ptr(g_fx),
                                         _ptr(w), _ptr(s), int(nb_steps), B, d, E, int(bool(inv_f)),
                                         _ptr(dx0), _ptr(dx), _ptr(dh), _ptr(dtheta), _ptr(ws), int(nbytes), stream)
    _lib.check(rc, "umnn_cc_backward")
    _state.path = _last_backward["path"] = "hip"
    if x_dtype != xd:
        dx0 = dx0.to(x_dtype) if dx0 is not None else None
        dx = dx.to(x_dtype) if dx is not None else None
    if h_dtype != hd and dh is not None:
        dh = dh.to(h_dtype)
    return dx0, dx, dh, dtheta


# ----------------------------------------------------------------------------------------------
# generic ATen path (arbitrary callables; any device)
# ----------------------------------------------------------------------------------------------
_CHUNK_ELEMS = 1 << 24      # cap on rows*columns of the largest temporary per chunk of nodes


def _node_chunks(nb_steps, rows, cols):
    per = max(1, min(nb_steps + 1, _CHUNK_ELEMS // max(1, rows * cols)))
    return [(a, min(a + per, nb_steps + 1)) for a in range(0, nb_steps + 1, per)]


def _eval_chunk(integrand, x0, span, h, u):
    """Evaluate the integrand at nodes t = x0 + span*u_c/2 for a chunk of C nodes -> ([C,B,dout], h_rep)."""
    C, B = u.shape[0], x0.shape[0]
    t = (x0.unsqueeze(0) + span.unsqueeze(0) * u.view(C, 1, 1) / 2).reshape(C * B, -1)
    h_rep = h.unsqueeze(0).expand(C, -1, -1).reshape(C * B, -1)
    return t, h_rep


def aten_forward(integrand, x0, x, h, nb_steps, inv_f=False):
    w, s = device_tables(nb_steps, x.device)
    w, u = w.to(x.dtype), s.to(x.dtype) + 1
    span = x - x0
    total = torch.zeros_like(x)
    B = x.shape[0]
    for a, b in _node_chunks(nb_steps, B, h.shape[1] + x.shape[1]):
        t, h_rep = _eval_chunk(integrand, x0, span, h, u[a:b])
        f = integrand(t, h_rep)
        if inv_f:
            f = 1 / f
        total = total + (f.view(b - a, B, -1) * w[a:b].view(-1, 1, 1)).sum(0)
    _state.path = "aten"
    return total * span / 2


def aten_backward(integrand, x0, x, h, g, nb_steps, inv_f=False):
    """d_theta (flat, parameters() order) and d_h: VJP of f over all nodes with cotangent g*(x-x0)/2*w_k."""
    w, s = device_tables(nb_steps, x.device)
    w, u = w.to(x.dtype), s.to(x.dtype) + 1
    span = x - x0
    cot = g * span / 2
    params = [p for p in integrand.parameters()] if isinstance(integrand, torch.nn.Module) else []
    g_params = [torch.zeros_like(p) for p in params]
    g_h = torch.zeros_like(h)
    B = x.shape[0]
    for a, b in _node_chunks(nb_steps, B, h.shape[1] + x.shape[1]):
        t, h_rep = _eval_chunk(integrand, x0, span, h, u[a:b])
        h_rep = h_rep.detach().requires_grad_(True)
        with torch.enable_grad():
            f = integrand(t.detach(), h_rep)
            if inv_f:
                f = 1 / f
            cot_c = (cot.unsqueeze(0) * w[a:b].view(-1, 1, 1)).reshape(f.shape)
            grads = torch.autograd.grad(f, params + [h_rep], cot_c, allow_unused=True)
        for acc, gr in zip(g_params, grads[:-1]):
            if gr is not None:
                acc += gr
        if grads[-1] is not None:
            g_h += grads[-1].view(b - a, B, -1).sum(0)
    _state.path = _last_backward["path"] = "aten"
    return (_flatten(g_params) if params else None), g_h


class IntegralWithJacobianParams(torch.autograd.Function):
    """IntegralWithJacobian with the integrand's parameters passed one by one instead of as one flat tensor: no
    ``torch.cat`` in the forward and no cat-backward (a narrow + copy per parameter) in the backward -- the gradients are
    views into the kernel's flat d_theta.  Internal to the flow blocks; the public operators keep the reference's
    ``flat_params`` signature."""

    @staticmethod
    def forward(ctx, x0, x, integrand, h, nb_steps, *params):
        spec = mlp_spec(integrand)
        if not _use_hip(spec, x):
            raise RuntimeError("IntegralWithJacobianParams needs an MLP integrand on a GPU")
        ctx.spec, ctx.nb_steps, ctx.integrand = spec, nb_steps, integrand
        ctx.shapes = [p.shape for p in params]
        ctx.x0_none = x0 is None                    # lower limit 0: no tensor to save, clone or differentiate
        if ctx.x0_none:
            ctx.save_for_backward(x.clone(), h)
        else:
            ctx.save_for_backward(x0.clone(), x.clone(), h)
        F, fx, _ = hip_forward(spec, x0, x, h, nb_steps, False)
        return F, fx

    @staticmethod
    @once_differentiable        # double backward (create_graph=True through the quadrature) raises instead of silently detaching
    def backward(ctx, gF, gfx):
        if ctx.x0_none:
            (x, h), x0 = ctx.saved_tensors, None
        else:
            x0, x, h = ctx.saved_tensors
        if not _hip_backward_ok(ctx.spec, x, h):
            dx0, dx, dh, dtheta = aten_backward_jac(ctx.integrand, torch.zeros_like(x) if x0 is None else x0, x, h, gF, gfx,
                                                    ctx.nb_steps)
        else:
            need = (ctx.needs_input_grad[0] and x0 is not None, ctx.needs_input_grad[1], ctx.needs_input_grad[3],
                    any(ctx.needs_input_grad[5:]))
            dx0, dx, dh, dtheta = hip_backward(ctx.spec, x0, x, h, gF, gfx, ctx.nb_steps, need)
        grads, o = [], 0
        for shp, needed in zip(ctx.shapes, ctx.needs_input_grad[5:]):
            n = int(torch.Size(shp).numel())
            grads.append(dtheta[o:o + n].view(shp) if (needed and dtheta is not None) else None)
            o += n
        return (None if ctx.x0_none else dx0, dx, None, dh, None, *grads)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FlowBlockTransform(torch.autograd.Function):
    """(z, log_jac) of one UMNN-MAF block on the TRAINING path as ONE autograd node (UMNNMAF.py:76-139):

        z[b, rev(i)] = exp(s_i) (int_0^{x_bi} f(t; h_b) dt + h_0[b,i]),     log_jac = [log_jac_in +] log(f(x; h) + 1e-10) + s

    forward = the fused-epilogue launch of the inference path (umnn_flow_stack_block_forward); backward = one elementwise launch
    (umnn_flow_block_cotangents: cotangents of F and f_x from those of z and log_jac), the quadrature backward, and the h_0 term
    added into d_h.  Composed from torch ops (exp, mul, add, log, add, flip, select and their backward nodes) the same
    arithmetic is ~15 launches and 8 autograd nodes per block.  Internal to the flow blocks: fp32 storage, lower limit 0, frozen
    ``scaling`` (UMNNMAF.py:53) -- anything else keeps the composed path."""

    @staticmethod
    def forward(ctx, x, integrand, h, scaling, nb_steps, reverse_z, log_jac_in, *params):
        spec = mlp_spec(integrand)
        ctx.spec, ctx.nb_steps, ctx.integrand, ctx.reverse_z = spec, nb_steps, integrand, bool(reverse_z)
        ctx.shapes = [p.shape for p in params]
        # (nets of the three-stage backward family: the forward leaves z_2 of every node for the backward, which then skips stage A)
        want_z2 = any(ctx.needs_input_grad[7:]) or ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        out = hip_flow_block(spec, x, h, scaling, nb_steps, reverse_z, log_jac_in, save_z2=want_z2)
        z, lj, fx = out[0], out[1], out[2]
        ctx.z2 = out[4] if want_z2 else None          # (save_z2=False returns four values: only log_jac_in needs a gradient)
        ctx.save_for_backward(x.clone(), h, fx, scaling)      # (x cloned: callers clamp z / reuse x in place, UMNNMAF.py:150)
        return z, lj

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, glj):
        x, h, fx, scaling = ctx.saved_tensors
        lib = _lib.lib()
        B, d = x.shape
        gz = None if gz is None else gz.contiguous()
        glj = None if glj is None else glj.contiguous()
        gF, gfx = torch.empty_like(x), (torch.empty_like(x) if glj is not None else None)
        with torch.cuda.device(x.device):
            rc = lib.umnn_flow_block_cotangents(_ptr(gz), _ptr(glj), _ptr(fx), _ptr(scaling), B, d, 1 if ctx.reverse_z else 0,
                                                _ptr(gF), _ptr(gfx), _stream(x.device))
        _lib.check(rc, "umnn_flow_block_cotangents")
        if not _hip_backward_ok(ctx.spec, x, h):
            _, dx, dh, dtheta = aten_backward_jac(ctx.integrand, torch.zeros_like(x), x, h, gF, gfx, ctx.nb_steps)
        else:
            need = (False, ctx.needs_input_grad[0], ctx.needs_input_grad[2], any(ctx.needs_input_grad[7:]))
            _, dx, dh, dtheta = hip_backward(ctx.spec, None, x, h, gF, gfx, ctx.nb_steps, need, z2_saved=ctx.z2)
            ctx.z2 = None
        if dh is not None:
            dh.view(B, -1, d)[:, 0, :].add_(gF)            # z carries h_0 = embedding row 0 (UMNNMAF.py:80)
        grads, o = [], 0
        for shp, needed in zip(ctx.shapes, ctx.needs_input_grad[7:]):
            n = int(torch.Size(shp).numel())
            grads.append(dtheta[o:o + n].view(shp) if (needed and dtheta is not None) else None)
            o += n
        return (dx, None, dh, None, None, None, glj if ctx.needs_input_grad[6] else None, *grads)


class FlowLogLikelihood(torch.autograd.Function):
    """ll[b] = sum_i log_jac[b,i] - 1/2 sum_i (log 2 pi + z[b,i]^2)  (UMNNMAFFlow.py:109-119) as one launch per direction."""

    @staticmethod
    def forward(ctx, z, log_jac):
        z, log_jac = z.contiguous(), log_jac.contiguous()
        B, d = z.shape
        ll = torch.empty(B, device=z.device, dtype=torch.float32)
        with torch.cuda.device(z.device):
            rc = _lib.lib().umnn_flow_ll_forward(_ptr(z), _ptr(log_jac), B, d, _ptr(ll), _stream(z.device))
        _lib.check(rc, "umnn_flow_ll_forward")
        ctx.save_for_backward(z)
        return ll

    @staticmethod
    @once_differentiable
    def backward(ctx, g_ll):
        (z,) = ctx.saved_tensors
        B, d = z.shape
        g_ll = g_ll.contiguous()
        gz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        glj = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(z.device):
            rc = _lib.lib().umnn_flow_ll_backward(_ptr(z), _ptr(g_ll), B, d, _ptr(gz), _ptr(glj), _stream(z.device))
        _lib.check(rc, "umnn_flow_ll_backward")
        return gz, glj


def fused_block_ok(x, h, scaling, x0, want_jac):
    """The one-node training path of a block applies: fp32 x-class storage (the embedding h in fp32 or bf16), lower limit 0, log_jac
    wanted, frozen scaling, no autocast."""
    return (x0 is None and want_jac and x.dtype == torch.float32 and h.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2
            and not scaling.requires_grad and not torch.is_autocast_enabled() and os.environ.get("UMNN_FUSED_TRAIN", "1") != "0")


# ----------------------------------------------------------------------------------------------
# reference-shaped public API
# ----------------------------------------------------------------------------------------------
def integrate(x0, nb_steps, step_sizes, integrand, h, compute_grad=False, x_tot=None, inv_f=False,
              cc_weights=None, steps=None):
    """Clenshaw-Curtis quadrature of ``integrand`` from x0 to x0 + nb_steps*step_sizes.

    compute_grad=False -> the integral [B,d].  compute_grad=True -> (d_theta_flat, d_h) for cotangent ``x_tot``
    (what the reference's backward consumes).  ``cc_weights``/``steps`` are accepted for signature parity; the
    tables are a pure function of nb_steps and come from the per-device cache."""
    x = x0 + nb_steps * step_sizes
    spec = mlp_spec(integrand)
    if not compute_grad:
        # The reference's direct integration is plain ATen, hence differentiable by ordinary autograd
        # (ParallelNeuralIntegral.py:49-65).  Keep that: only when nothing can ask for a gradient is the graph skipped.
        wants_graph = torch.is_grad_enabled() and (
            x.requires_grad or (h is not None and h.requires_grad)
            or (isinstance(integrand, torch.nn.Module) and any(p.requires_grad for p in integrand.parameters())))
        if _use_hip(spec, x):
            if wants_graph:
                return ParallelNeuralIntegral.apply(x0, x, integrand, _flatten(integrand.parameters()), h, nb_steps, inv_f)
            return hip_forward(spec, x0, x, h, nb_steps, inv_f)[0]
        if wants_graph:
            return aten_forward(integrand, x0, x, h, nb_steps, inv_f)
        with torch.no_grad():
            return aten_forward(integrand, x0, x, h, nb_steps, inv_f)
    if _use_hip(spec, x) and _hip_backward_ok(spec, x, h):
        _, _, dh, dtheta = hip_backward(spec, x0, x, h, x_tot, None, nb_steps, need=(False, False, True, True), inv_f=inv_f)
        return dtheta, dh
    return aten_backward(integrand, x0, x, h, x_tot, nb_steps, inv_f)


def _op_forward(ctx, x0, x, integrand, h, nb_steps, inv_f):
    ctx.integrand, ctx.nb_steps, ctx.inv_f = integrand, nb_steps, inv_f
    spec = mlp_spec(integrand)
    ctx.spec = spec
    # clones: callers mutate their tensors in place after the call (UMNNMAF.compute_ll clamps z, :150)
    ctx.save_for_backward(x0.clone(), x.clone(), h)
    ctx.use_hip = _use_hip(spec, x)       # decided on the calling thread (force_generic is thread-local; backward runs elsewhere)
    if ctx.use_hip:
        return hip_forward(spec, x0, x, h, nb_steps, inv_f)[0]
    return aten_forward(integrand, x0, x, h, nb_steps, inv_f)


def _op_backward(ctx, grad_output):
    x0, x, h = ctx.saved_tensors
    integrand, nb_steps, inv_f, spec = ctx.integrand, ctx.nb_steps, ctx.inv_f, ctx.spec
    if ctx.use_hip and _hip_backward_ok(spec, x, h):
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[4], ctx.needs_input_grad[3])
        dx0, dx, dh, dtheta = hip_backward(spec, x0, x, h, grad_output, None, nb_steps, need, inv_f=inv_f)
        return dx0, dx, dtheta, dh
    dtheta, dh = aten_backward(integrand, x0, x, h, grad_output, nb_steps, inv_f)
    with torch.no_grad():
        dx = integrand(x, h) * grad_output
        dx0 = -integrand(x0, h) * grad_output
    return dx0, dx, dtheta, dh.view(h.shape)


class ParallelNeuralIntegral(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x, integrand, flat_params, h, nb_steps=20, inv_f=False):
        return _op_forward(ctx, x0, x, integrand, h, nb_steps, inv_f)

    @staticmethod
    @once_differentiable        # double backward (create_graph=True through the quadrature) raises instead of silently detaching
    def backward(ctx, grad_output):
        dx0, dx, dtheta, dh = _op_backward(ctx, grad_output)
        return dx0, dx, None, dtheta, dh, None, None


class NeuralIntegral(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x, integrand, flat_params, h, nb_steps=20):
        return _op_forward(ctx, x0, x, integrand, h, nb_steps, False)

    @staticmethod
    @once_differentiable        # double backward (create_graph=True through the quadrature) raises instead of silently detaching
    def backward(ctx, grad_output):
        dx0, dx, dtheta, dh = _op_backward(ctx, grad_output)
        return dx0, dx, None, dtheta, dh, None


class IntegralWithJacobian(torch.autograd.Function):
    """(F, f_x) = (int_{x0}^{x} f, f(x;h)) in one kernel pass, differentiable in both outputs.

    This is what a UMNNMAF block needs (z from F, log-det from f_x); the reference obtains f_x from a second
    integrand evaluation and a second MADE pass (UMNNMAF.py:136-139).  HIP path only."""

    @staticmethod
    def forward(ctx, x0, x, integrand, flat_params, h, nb_steps):
        spec = mlp_spec(integrand)
        if not _use_hip(spec, x):
            raise RuntimeError("IntegralWithJacobian needs an MLP integrand on a GPU")
        ctx.spec, ctx.nb_steps, ctx.integrand = spec, nb_steps, integrand
        ctx.save_for_backward(x0.clone(), x.clone(), h)
        F, fx, _ = hip_forward(spec, x0, x, h, nb_steps, False)
        return F, fx

    @staticmethod
    @once_differentiable        # double backward (create_graph=True through the quadrature) raises instead of silently detaching
    def backward(ctx, gF, gfx):
        x0, x, h = ctx.saved_tensors
        if not _hip_backward_ok(ctx.spec, x, h):
            dx0, dx, dh, dtheta = aten_backward_jac(ctx.integrand, x0, x, h, gF, gfx, ctx.nb_steps)
            return dx0, dx, None, dtheta, dh, None
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[4], ctx.needs_input_grad[3])
        dx0, dx, dh, dtheta = hip_backward(ctx.spec, x0, x, h, gF, gfx, ctx.nb_steps, need)
        return dx0, dx, None, dtheta, dh, None
