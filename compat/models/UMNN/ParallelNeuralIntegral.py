"""Shim for `models.UMNN.ParallelNeuralIntegral` -> umnn_amd.integral."""
from umnn_amd.integral import *  # noqa: F401,F403
from umnn_amd import integral as _impl
from umnn_amd.quadrature import compute_cc_weights  # noqa: F401
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
