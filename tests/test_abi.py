"""The C-ABI library loads on a CPU-only box and exports every symbol include/umnn_cc.h declares; argument
validation and the host-side helpers work without a GPU (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import _util as U
from umnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "umnn_cc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(umnn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 10
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/umnn_cc.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in umnn_amd/_lib.py"
    assert _lib.lib().umnn_version() >= 100


def _desc(widths, ptr=0x1000):
    d = _lib.MlpDesc()
    d.n_linear = len(widths) - 1
    for i, w in enumerate(widths):
        d.widths[i] = w
    for l in range(d.n_linear):
        d.W[l], d.b[l] = ptr, ptr
    return d


def test_flops_per_integral_matches_survey():
    lib = _lib.lib()
    uci = _desc([31, 50, 50, 50, 50, 1])
    assert lib.umnn_cc_forward_flops_per_integral(ctypes.byref(uci), 100) == 2 * (101 * 7600 + 1500)   # 1.538 MFLOP
    assert lib.umnn_cc_forward_flops_per_integral(ctypes.byref(uci), 50) == 2 * (51 * 7600 + 1500)     # 0.778 MFLOP
    toy = _desc([11, 100, 100, 100, 100, 1])
    assert lib.umnn_cc_forward_flops_per_integral(ctypes.byref(toy), 50) == 2 * (51 * 30200 + 1000)    # 3.082 MFLOP


@pytest.mark.parametrize("n", [5, 7, 20, 30, 50, 51, 100, 200])
def test_tables_host_match_reference(n):
    G = U.load("g1_cc_tables")
    w = (ctypes.c_float * (n + 1))()
    s = (ctypes.c_float * (n + 1))()
    assert _lib.lib().umnn_cc_tables_host(n, w, s) == 0
    # C loop vs numpy matmul sum in float64: identical after the cast except for rare last-bit ties
    np.testing.assert_allclose(np.array(w), G[f"w{n}"].reshape(-1), rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.array(s), G[f"s{n}"].reshape(-1), rtol=0, atol=6e-8)


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    p = ctypes.c_void_p(0x1000)
    bad_in = _desc([30, 50, 50, 1])          # widths[0] != 1+E
    rc = lib.umnn_cc_forward(ctypes.byref(bad_in), None, p, p, p, p, 20, 4, 2, 30, 0, p, None, None, None)
    assert rc == -1 and b"widths[0]" in lib.umnn_last_error()
    too_wide = _desc([3, 128, 1])
    rc = lib.umnn_cc_forward(ctypes.byref(too_wide), None, p, p, p, p, 20, 4, 2, 2, 0, p, None, None, None)
    assert rc == -2
    no_hidden = _desc([3, 1])
    rc = lib.umnn_cc_forward(ctypes.byref(no_hidden), None, p, p, p, p, 20, 4, 2, 2, 0, p, None, None, None)
    assert rc == -2
    ok = _desc([3, 16, 1])
    rc = lib.umnn_cc_forward(ctypes.byref(ok), None, None, p, p, p, 20, 4, 2, 2, 0, p, None, None, None)
    assert rc == -1 and b"non-null" in lib.umnn_last_error()
    rc = lib.umnn_cc_forward(ctypes.byref(ok), None, p, p, p, p, 0, 4, 2, 2, 0, p, None, None, None)
    assert rc == -1
    # empty batch is a no-op, not an error (and touches no device)
    rc = lib.umnn_cc_forward(ctypes.byref(ok), None, p, p, p, p, 20, 0, 2, 2, 0, p, None, None, None)
    assert rc == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_set_option_validates_ranges_without_gpu():
    """umnn_set_option accepts exactly the values load_env accepts: anything else is UMNN_EINVAL and leaves
    the option untouched; the documented values round-trip."""
    for name, good, bad in (("fwd_p", (1, 2, -1), (0, 3)), ("fwd_ns", (1, 2, 4, -1), (3, 8)), ("fwd_precision", (0, 1, 2, 3), (-1, 4, 7)),
                            ("bwd_precision", (0, 1), (2,)), ("bwd_ns", (1, 32, -1), (0, 33, 64)), ("fwd_pipe", (0, 1, 2), (3, -1)),
                            ("bwd_swp", (0, 1), (2, -1)), ("fwd_tail", (0, 1, -1), (2,)), ("bwd_ws16", (0, 1, 2), (3, -1))):
        old = _lib.get_option(name)
        try:
            for v in good:
                _lib.set_option(name, v)
                assert _lib.get_option(name) == v
            for v in bad:
                before = _lib.get_option(name)
                with pytest.raises(RuntimeError):
                    _lib.set_option(name, v)
                assert _lib.get_option(name) == before
        finally:
            _lib.set_option(name, old)
    with pytest.raises(RuntimeError):
        _lib.set_option("no_such_option", 1)
