"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/blasted_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from blasted_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "blasted_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(blasted_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = header_symbols()
    assert len(names) >= 25
    assert sorted(capi.SYMBOLS) == names
    lib = capi.lib()
    for n in names:
        assert getattr(lib, n) is not None


def test_no_torch_or_cxx_types_in_header():
    txt = open(os.path.join(ROOT, "include", "blasted_hip.h")).read()
    assert 'extern "C"' in txt
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert "torch" not in txt and "std::" not in txt and "template" not in txt


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "blasted_hip.h"\nint main(void){return BLASTED_HIP_OK;}\n')
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert capi.device_count() == 0
    with pytest.raises(capi.BlastedHipError) as ei:
        capi.Prec(0)
    assert ei.value.code == capi.ENODEV
    assert "no CPU fallback" in str(ei.value)


def test_product_never_touches_the_oracle():
    """Nothing under blasted_amd/ or include/ may import, link or call oracle/."""
    bad = []
    for base in ("blasted_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in dp or dp.endswith("lib"):
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", "Makefile")):
                    s = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle\b|orc_|blasted_oracle", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
