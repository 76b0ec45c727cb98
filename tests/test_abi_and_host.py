"""CPU-side checks (no GPU): the C-ABI library loads and exports exactly what include/rsu.h declares; host logic
(input_size_needed, parameter table, lr schedule, flags) matches the reference's behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "rsu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsu_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from road_segmentation_unet_amd import _lib
    L = _lib.lib()  # raises if the .so is missing: there is no fallback path
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), "librsu_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header out of sync"
    assert b"gfx950" in L.rsu_version()


def test_missing_extension_fails_loudly(monkeypatch):
    from road_segmentation_unet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librsu_hip.so")
    with pytest.raises(_lib.RsuError):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "road_segmentation_unet_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.lower().replace("test oracle", ""), "product file %s mentions the oracle" % fn


def test_input_size_needed_abi_and_python(golden):
    from road_segmentation_unet_amd import _lib, input_size_needed
    L = _lib.lib()
    for nl, P, S in golden["g6_table"]:
        out = ctypes.c_int(-1)
        rc = L.rsu_input_size_needed(int(P), int(nl), ctypes.byref(out))
        if S < 0:
            assert rc == -22
            with pytest.raises(AssertionError):
                input_size_needed(int(P), int(nl))
        else:
            assert rc == 0 and out.value == S == input_size_needed(int(P), int(nl))
    with pytest.raises(AssertionError) as ei:
        input_size_needed(128, 5)
    assert str(ei.value) == str(golden["g6_assert_msg"])  # the reference's own assertion text (unet.py:108)


def test_param_table_matches_oracle_and_reference_counts():
    from oracle import unet_oracle as U
    from road_segmentation_unet_amd.unet import param_shapes
    for L, root, dil in [(5, 64, False), (6, 64, True), (3, 16, False), (4, 8, True)]:
        assert param_shapes(L, root, dil) == U.param_shapes(L, root, dil)
    assert sum(int(np.prod(s)) for _, s in param_shapes(6, 64, True)) == 212403278  # report/report.tex:50 "2e8"
    assert sum(int(np.prod(s)) for _, s in param_shapes(5, 64, False)) == 31031822


def test_packed_bytes_formula():
    from road_segmentation_unet_amd import _lib
    seg = (ctypes.c_int * 3)(64, 64, 64)
    assert _lib.lib().rsu_packed_bytes(9, 64, seg, 3) == 9 * 192 * 128 * 2
    seg = (ctypes.c_int * 2)(16, 16)
    assert _lib.lib().rsu_packed_bytes(9, 16, seg, 2) == 9 * 64 * 128 * 2
