"""The C-ABI library loads on a machine without a GPU and exports every entry
point include/epi.h declares (no device call is made here)."""
import ctypes as C
import os
import re

import pytest

from epidemicmodeling_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="epi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epi_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_testing_build_has_the_switches_and_the_product_does_not():
    assert _declared("epi_testing.h") == sorted(_lib.TESTING_SIGNATURES)
    product, testing = _lib.load(), _lib.load_testing()
    for name in _lib.TESTING_SIGNATURES:
        assert hasattr(testing, name) and not hasattr(product, name), name
    for name in _lib.SIGNATURES:
        assert hasattr(testing, name), name


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.epi_state_rows(14, 0) == 16 and lib.epi_state_rows(14, 5) == 2 + 14 + 5 * 13
    assert lib.epi_pf_frac_bits(1000) == 40 and lib.epi_pf_frac_bits(10 ** 8) == 26 and lib.epi_pf_frac_bits(10 ** 6) == 33
    # plan: equal weights -> every shard owns its own slots; all weight in shard 1 -> it owns everything
    w = (C.c_uint64 * 2)(1000, 1000)
    lo = (C.c_int64 * 3)()
    assert lib.epi_pf_plan(100, 2, w, 5, 2, lo) == 0 and list(lo) == [0, 50, 100]
    w = (C.c_uint64 * 2)(0, 1000)
    assert lib.epi_pf_plan(100, 2, w, 5, 2, lo) == 0 and list(lo) == [0, 0, 100]
    w = (C.c_uint64 * 2)(0, 0)
    assert lib.epi_pf_plan(101, 2, w, 5, 2, lo) == 0 and list(lo) == [0, 50, 101]
    assert lib.epi_pf_plan(0, 2, w, 5, 2, lo) == -1 and b"epi_pf_plan" in lib.epi_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libepi_gfx950.so")
    with pytest.raises(_lib.EpiLibraryError, match="no CPU fallback"):
        _lib.load()
