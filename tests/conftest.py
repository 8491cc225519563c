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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()
