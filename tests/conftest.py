import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle's OpenMP regions are entered thousands of times on tiny matrices; with one thread per
# hardware thread of a large host (the GPU boxes report 128 but grant a share of them) the fork/join
# barriers dominate.  A modest team keeps the checker fast; bench.py's cpu_baseline is not affected.
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# The product makes its compact triangle copies when the 16th sweep application since a factorisation comes along
# (capi.hip, g_compact_after); the tests apply an operator a few times each and are meant to cover the kernels ON the
# copies: here the copies are made with the first application (test_gpu_parity.py::test_compact_copies_are_made_lazily
# covers the policy itself).  Set before the library is loaded, inherited by the native drivers.
os.environ.setdefault("BLASTED_HIP_COMPACT_AFTER", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()
