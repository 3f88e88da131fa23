"""Shim for `models.UMNN.made` -> umnn_amd.made."""
from umnn_amd.made import *  # noqa: F401,F403
from umnn_amd import made as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
