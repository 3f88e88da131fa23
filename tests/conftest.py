import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def opts():
    """Set libumnn_cc launch options for the duration of a test: ``opts(fwd_p=2, fwd_ns=1)`` (restored afterwards)."""
    from umnn_amd import _lib
    saved = {}

    def setter(**kw):
        for k, v in kw.items():
            saved.setdefault(k, _lib.get_option(k))
            _lib.set_option(k, v)
    yield setter
    for k, v in saved.items():
        _lib.set_option(k, v)
