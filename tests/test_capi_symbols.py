"""CPU: the C-ABI library loads and exports exactly what include/orphics_amd.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "orphics_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from orphics_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(_lib.SIGNATURES.keys()) == syms  # ctypes table mirrors the header 1:1
    assert lib.oa_version() == _lib.ABI_VERSION


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from orphics_amd import _lib, maps
    from orphics_amd.geometry import FlatGeometry
    import numpy as np
    fc = maps.FourierCalc((64, 64), FlatGeometry.from_res((64, 64), 2.0))
    with pytest.raises(_lib.OrphicsAmdError):
        fc.fft(np.zeros((64, 64)))
    import ctypes
    h = ctypes.c_void_p()
    assert _lib.load().oa_plan_create(64, 64, 0, ctypes.byref(h)) != 0
    assert b"device" in _lib.load().oa_last_error().lower() or b"hip" in _lib.load().oa_last_error().lower()
