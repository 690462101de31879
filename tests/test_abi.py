"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/wah.h
declares plus the reference's two C++-linkage symbols.  No compute call is made (no GPU here)."""
import ctypes
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    p = importlib.import_module("gpu-wah_amd")
    p.build()
    return p


def test_library_loads_and_exports_header_symbols(pkg):
    lib = pkg.lib()
    header = open(os.path.join(ROOT, "include", "wah.h")).read()
    declared = set(re.findall(r"\b(wah_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found in include/wah.h"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/wah.h but not exported"
    assert declared == set(pkg.ABI_SYMBOLS), (declared ^ set(pkg.ABI_SYMBOLS))


def test_reference_cxx_symbols_exported(pkg):
    """compress.h:12-18 / decompress.h:11-17 have C++ linkage: the mangled names must exist."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.lib_path()], text=True)
    for sym in ("_Z8compressPjyPyPfS1_S1_", "_Z10decompressPjyPyPfS1_S1_"):
        assert re.search(rf"\bT {sym}\b", out), sym


def test_nothing_else_is_exported(pkg):
    """The dynamic symbol table holds the extern "C" ABI of include/wah.h and the reference's two C++ entry points --
    no launcher, no layout helper, no kernel stub, no template instantiation of the C++ runtime (exports.map)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.lib_path()], text=True)
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    allowed = set(pkg.ABI_SYMBOLS) | {"_Z8compressPjyPyPfS1_S1_", "_Z10decompressPjyPyPfS1_S1_"}
    assert names == allowed, sorted(names ^ allowed)


def test_size_helpers_match_reference_formulas(pkg, oracle):
    lib = pkg.lib()
    for n in (0, 1, 30, 31, 32, 992, 993, 262144, 268435200, 268435456):
        assert lib.wah_max_compressed_words(n) == oracle.max_words(n) == (32 * n + 30) // 31  # compress.cu:74-81
    for g in (0, 1, 31, 32, 33, 1024, 277094665):
        assert lib.wah_decoded_words(g) == oracle.decoded_words(g) == (31 * g + 31) // 32  # decompress.cu:84-93
    assert lib.wah_compress_workspace_bytes(268435456) % 256 == 0
    assert lib.wah_decompress_workspace_bytes(1 << 20, 1 << 20) % 256 == 0
    assert b"gfx950" in lib.wah_version()


def test_no_gpu_fails_loudly(pkg):
    """Without a device the operators raise instead of silently computing on the CPU."""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.WahError):
        pkg.compress(np.zeros(992, np.uint32))
    with pytest.raises(pkg.WahError):
        pkg.DeviceCompressor(992)


def test_product_does_not_touch_the_oracle():
    """The product tree must not include, import, link or call anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "gpu-wah_amd")
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)
    out = subprocess.check_output(["ldd", os.path.join(pkg_dir, "libwah_hip.so")], text=True)
    assert "oracle" not in out
