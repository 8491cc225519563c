"""Randomised differential test of every operator and mode against the oracle (tools/fuzz_parity.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity(seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["fuzz_parity.py", "60", str(seed)]
        mod.main()
    finally:
        sys.argv = argv
