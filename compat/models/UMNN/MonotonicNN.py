"""Shim for `models.UMNN.MonotonicNN` -> umnn_amd.monotonic."""
from umnn_amd.monotonic import *  # noqa: F401,F403
from umnn_amd import monotonic as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
