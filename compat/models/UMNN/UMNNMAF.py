"""Shim for `models.UMNN.UMNNMAF` -> umnn_amd.flow."""
from umnn_amd.flow import *  # noqa: F401,F403
from umnn_amd import flow as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
