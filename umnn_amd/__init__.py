"""umnn_amd -- MI355X-native (gfx950) neural-integration hot path of UMNN behind the reference's module API.

Import surface mirrors models/UMNN/__init__.py:1-6 of the reference.
"""
import os as _os
import sys as _sys

# RCCL / device-tensor sharing between the ranks of a node goes through dmabuf IPC on this driver stack; the legacy mode fails with
# "hipIpcGetMemHandle: invalid argument".  The HSA runtime reads the variable ONCE, when the process first touches the GPU, so the
# default has to be in the environment before that -- i.e. here, at import, and BEFORE anything below can make a HIP call (an
# explicit setting wins).  Importing this package after the GPU is already initialised cannot change the mode any more: say so
# instead of failing later inside a collective.  The probe is torch.cuda.is_initialized() only -- Python state, no HIP call
# (torch.cuda.is_available() goes through hipGetDeviceCount, which may itself start the runtime and freeze the legacy mode).
if "HSA_ENABLE_IPC_MODE_LEGACY" not in _os.environ:
    _os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    _torch = _sys.modules.get("torch")
    if _torch is not None and _torch.cuda.is_initialized():
        import warnings as _warnings
        _warnings.warn("umnn_amd: HSA_ENABLE_IPC_MODE_LEGACY was not set when this process initialised the GPU; multi-process RCCL "
                       "(torch.distributed backend 'nccl') may fail with 'hipIpcGetMemHandle: invalid argument'.  Export "
                       "HSA_ENABLE_IPC_MODE_LEGACY=0 or import umnn_amd before the first CUDA call.", RuntimeWarning)

from .flow import UMNNMAFFlow, UMNNMAF, EmbeddingNetwork, IntegrandNetwork, ListModule
from .monotonic import MonotonicNN, IntegrandNN
from .made import MADE, ConditionnalMADE, MaskedLinear, invalidate_caches, set_made_fast_path, get_made_fast_path, set_made_fused
from .integral import NeuralIntegral, ParallelNeuralIntegral, IntegralWithJacobian, integrate, path_taken, backward_path_taken, set_backward_wide
from .nets import compute_lipschitz_linear
from .quadrature import compute_cc_weights
from .graphs import GraphedLL, GraphedTrainStep
from ._lib import set_forward_precision, get_forward_precision, set_backward_precision, get_backward_precision



def set_precision(name):
    """One switch for the arithmetic of the whole path.

    'f16x3'  : the library default -- hidden GEMMs of the forward kernels on TWO fp16 pieces / 3 cross terms: fp32-level accuracy
               (~5e-7 on F against float64; the exact-fp32 kernels: ~4e-7) at the speed of 'bf16x3'.  fp16's exponent range is
               handled, not assumed: a tile group in which a hidden activation leaves +-65504 is detected in its quadrature sum,
               writes nothing, and is recomputed by the bf16x3 build of the same kernel queued right behind the launch (idle cost:
               one scalar load per workgroup) -- those integrals come back at bf16x3 accuracy, never NaN, never a wrong number.
               The BOTTOM of fp16's range is not guarded: the low piece of a hidden activation below ~1e-3 is an fp16 subnormal
               (it vanishes below ~6e-5), so a net whose hidden activations are uniformly tiny loses relative accuracy on the part of
               f the hidden layers carry -- measured (tests/test_gpu_round6.py, activations of order 1e-2 / 1e-3 / 1e-4): 1.1e-5 /
               1.3e-4 / 1.5e-3 of that part, against 5.6e-6 / 5.4e-5 / 5.5e-4 for exact fp32 arithmetic; F and f(x) themselves stay
               at 2e-7 of max(|ref|, 1).  Use 'bf16x6' or 'fp32' for such nets.
               Conditioner inference GEMMs as K-concatenated bf16 GEMMs (3e-6 of the output range).
    'fp32'   : exact fp32 products everywhere (fp32 MFMA kernels forward and backward, fp32 conditioner GEMMs) -- the
               reference's arithmetic, ~2.4x slower forward.
    'bf16x6' : hidden GEMMs on the bf16 matrix cores with 6 cross terms (fp32-level accuracy, ~4e-7 on F).
    'bf16x3' : 3 cross terms on bf16 pieces (F to ~6e-6, every parity test passes at 1e-4) -- the default until round 4, and the
               arithmetic of the overflow fallback.
    The backward kernels know 'fp32' and 'bf16x3' (fp32-level recompute + 3-term delta / dW; on fp16 pieces at large batch);
    'bf16x6' and 'f16x3' select 'bf16x3' there."""
    if name not in ("fp32", "bf16x3", "bf16x6", "f16x3"):
        raise ValueError(name)
    set_forward_precision(name)
    set_backward_precision("fp32" if name == "fp32" else "bf16x3")
    set_made_fast_path(name != "fp32")


__all__ = ["UMNNMAFFlow", "UMNNMAF", "EmbeddingNetwork", "IntegrandNetwork", "ListModule", "MonotonicNN",
           "IntegrandNN", "MADE", "ConditionnalMADE", "MaskedLinear", "NeuralIntegral", "ParallelNeuralIntegral",
           "IntegralWithJacobian", "integrate", "compute_cc_weights", "path_taken", "GraphedLL", "GraphedTrainStep",
           "set_precision", "invalidate_caches", "set_made_fast_path", "get_made_fast_path", "set_forward_precision", "get_forward_precision", "set_backward_precision", "get_backward_precision",
           "set_backward_wide", "compute_lipschitz_linear", "backward_path_taken", "set_made_fused"]
