import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; GPU tests fail loudly (not skip) when it is missing."""
    from nnuzoo_amd import _lib
    return _lib.load()


@pytest.fixture
def force_scan_gen2():
    """Context-manager factory: inside `with force_scan_gen2(clb):` every selective-scan / cross-scan call that generation
    2 (channels on the lanes, csrc/ss2d_scan_rl.hpp) CAN take does take it, whatever its size (by default only calls of
    >= 2 M row-steps do - the bench shapes, which no golden reaches); yields a callable returning the number of calls that
    took the generation-2 kernels since entry."""
    import contextlib

    @contextlib.contextmanager
    def _force(clb: int = 0):
        from nnuzoo_amd._lib import call, load
        lib = load()
        saved = [lib.nnz_scan_tuning_get(k) for k in range(3)]
        call("nnz_scan_tuning", 0, 1)
        call("nnz_scan_tuning", 1, clb)
        call("nnz_scan_tuning", 2, 0)
        before = lib.nnz_scan_tuning_get(3)
        try:
            yield lambda: lib.nnz_scan_tuning_get(3) - before
        finally:
            for k, v in enumerate(saved):
                call("nnz_scan_tuning", k, v)
    return _force
