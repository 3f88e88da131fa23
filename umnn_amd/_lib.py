"""ctypes binding of libumnn_cc.so (the C ABI declared in include/umnn_cc.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950).  There is
no fallback: if it cannot be loaded, every HIP-path call raises.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UMNN_CC_LIB") or os.path.join(_HERE, "libumnn_cc.so")     # (override: A/B runs of two builds)

MAX_LINEAR = 8
EINVAL, EUNSUPPORTED, ENODEVICE = -1, -2, -3
ACT_LEAKY_RELU, ACT_RELU = 0, 1
OUT_ELU_PLUS_ONE, OUT_SIGMOID = 0, 1

_fp = ctypes.c_void_p          # device pointers travel as integers
_ll = ctypes.c_longlong


class MlpDesc(ctypes.Structure):
    """struct umnn_mlp"""
    _fields_ = [
        ("n_linear", ctypes.c_int),
        ("widths", ctypes.c_int * (MAX_LINEAR + 1)),
        ("W", ctypes.c_void_p * MAX_LINEAR),
        ("b", ctypes.c_void_p * MAX_LINEAR),
        ("hidden_act", ctypes.c_int),
        ("out_act", ctypes.c_int),
    ]


class IoDesc(ctypes.Structure):
    """struct umnn_io: storage dtype of the x-class tensors and of h (0 fp32, 1 bf16)"""
    _fields_ = [("x_dtype", ctypes.c_int), ("h_dtype", ctypes.c_int)]


DTYPE_F32, DTYPE_BF16 = 0, 1

# name -> (restype, argtypes); must list every symbol include/umnn_cc.h declares
SIGNATURES = {
    "umnn_cc_forward": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                       _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]),
    "umnn_flow_block_forward": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                               _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]),
    "umnn_flow_stack_block_forward": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                                     _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                                     _fp, _fp, _fp, _fp, _fp]),
    "umnn_cc_forward_io": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(IoDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                          _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]),
    "umnn_flow_stack_block_forward_io": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(IoDesc), _fp, _fp, _fp, _fp, _fp,
                                                        ctypes.c_int, _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                                        _fp, _fp, _fp, _fp, _fp]),
    "umnn_cc_backward_io": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(IoDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                           ctypes.c_int, _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _ll, _fp]),
    "umnn_flow_ll_block_forward": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                                  _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  _fp, _fp, _fp, _fp, _fp]),
    "umnn_flow_invert_dim": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                            _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    "umnn_cc_backward": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                        _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _ll, _fp]),
    "umnn_cc_backward_workspace_bytes": (_ll, [ctypes.POINTER(MlpDesc), _ll, ctypes.c_int, ctypes.c_int]),
    "umnn_cc_tables_host": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                           ctypes.POINTER(ctypes.c_float)]),
    "umnn_cc_forward_flops_per_integral": (ctypes.c_double, [ctypes.POINTER(MlpDesc), ctypes.c_int]),
    "umnn_last_error": (ctypes.c_char_p, []),
    "umnn_version": (ctypes.c_int, []),
    "umnn_launch_count": (_ll, []),
    "umnn_last_kernel_name": (ctypes.c_char_p, []),
    "umnn_cc_forward_timed": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                             _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_float), _fp]),
    "umnn_set_forward_precision": (ctypes.c_int, [ctypes.c_int]),
    "umnn_get_forward_precision": (ctypes.c_int, []),
    "umnn_cc_backward_kind": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.c_int]),
    "umnn_set_backward_precision": (ctypes.c_int, [ctypes.c_int]),
    "umnn_get_backward_precision": (ctypes.c_int, []),
    "umnn_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "umnn_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_ll),
                                         ctypes.POINTER(ctypes.c_double)]),
    "umnn_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "umnn_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "umnn_reload_env": (ctypes.c_int, []),
    "umnn_profile_read_tag": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_ll),
                                             ctypes.POINTER(ctypes.c_double)]),
    "umnn_last_kernel_name_of": (ctypes.c_char_p, [ctypes.c_int]),
    "umnn_cc_forward_z2_floats": (_ll, [ctypes.POINTER(MlpDesc), _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "umnn_flow_stack_block_forward_save": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                                          _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                                          _fp, _fp, _fp, _fp, _fp, _ll, _fp]),
    "umnn_cc_backward_saved": (ctypes.c_int, [ctypes.POINTER(MlpDesc), _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                              _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _ll, _fp, _ll, _fp]),
    "umnn_flow_stack_block_forward_save_io": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(IoDesc), _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                                             _ll, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                                             _fp, _fp, _fp, _fp, _fp, _ll, _fp]),
    "umnn_cc_backward_saved_io": (ctypes.c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(IoDesc), _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                                 _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _ll, _fp, _ll, _fp]),
    "umnn_flow_block_cotangents": (ctypes.c_int, [_fp, _fp, _fp, _fp, _ll, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    "umnn_flow_ll_forward": (ctypes.c_int, [_fp, _fp, _ll, ctypes.c_int, _fp, _fp]),
    "umnn_flow_ll_backward": (ctypes.c_int, [_fp, _fp, _ll, ctypes.c_int, _fp, _fp, _fp]),
    "umnn_made_split3": (ctypes.c_int, [_fp, _ll, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp]),
    "umnn_made_launch_count": (ctypes.c_longlong, []),
    "umnn_made_relu_bwd_bias_row_blocks": (ctypes.c_int, [_ll, ctypes.c_int]),
    "umnn_made_relu_bwd_bias": (ctypes.c_int, [_fp, _fp, _ll, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp]),
    "umnn_last_made_kernel_name": (ctypes.c_char_p, []),
    "umnn_made_mlp_forward": (ctypes.c_int, [ctypes.c_void_p, _fp, _ll, _fp, ctypes.c_int, _fp]),
    "umnn_made_mlp_forward_ex": (ctypes.c_int, [ctypes.c_void_p, _fp, _ll, _fp, ctypes.c_int, ctypes.c_int, _fp]),
    "umnn_made_linear_forward": (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int, _ll, ctypes.c_int, _fp,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
}

MADE_MAX_LAYERS = 8


class MadeNet(ctypes.Structure):
    """struct umnn_made_net (include/umnn_cc.h)."""
    _fields_ = [("n_layers", ctypes.c_int), ("widths", ctypes.c_int * (MADE_MAX_LAYERS + 1)),
                ("W", ctypes.c_void_p * MADE_MAX_LAYERS), ("b", ctypes.c_void_p * MADE_MAX_LAYERS)]

PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}
PROF_FORWARD, PROF_BACKWARD, PROF_FINISH = 0, 1, 2


def profile_read(tag=None):
    """(kernel milliseconds, launches, algorithmic FLOPs) of the launches recorded since umnn_profile_enable(1)."""
    ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
    if tag is None:
        check(lib().umnn_profile_read(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "umnn_profile_read")
    else:
        check(lib().umnn_profile_read_tag(int(tag), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "umnn_profile_read_tag")
    return ms.value, n.value, fl.value

_lib = None
_lock = threading.Lock()


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises HipLibraryMissing loudly if absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HipLibraryMissing(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  umnn_amd has no non-HIP path for MLP integrands on GPU.")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().umnn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def set_forward_precision(name):
    """'f16x3' (default: two fp16 pieces, three cross terms -- fp32-level accuracy at the cost of bf16x3; tile groups whose pieces
    overflow fp16's range are recomputed by the queued bf16x3 build), 'fp32' (exact fp32 MFMA), 'bf16x3' (bf16 split, 3 cross terms)
    or 'bf16x6' (three bf16 pieces: fp32-level accuracy)."""
    check(lib().umnn_set_forward_precision(PRECISIONS[name]), "umnn_set_forward_precision")


def get_forward_precision():
    mode = lib().umnn_get_forward_precision()
    return next(k for k, v in PRECISIONS.items() if v == mode)


def set_backward_precision(name):
    """'fp32' or 'bf16x3' (default) for the GEMMs of the backward kernels."""
    check(lib().umnn_set_backward_precision(PRECISIONS[name]), "umnn_set_backward_precision")


def get_backward_precision():
    mode = lib().umnn_get_backward_precision()
    return next(k for k, v in PRECISIONS.items() if v == mode)


def set_option(name, value):
    """Launch option by name (include/umnn_cc.h: fwd_p, fwd_ns, fwd_tail, fwd_pipe, fwd_pad, bwd_ns, ...; -1 = automatic)."""
    check(lib().umnn_set_option(name.encode(), int(value)), "umnn_set_option")


def get_option(name):
    v = ctypes.c_int()
    check(lib().umnn_get_option(name.encode(), ctypes.byref(v)), "umnn_get_option")
    return v.value


class options:
    """Context manager: ``with _lib.options(fwd_p=2, fwd_ns=1): ...`` sets launch options and restores them on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
