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


def test_host_side_abi_rules_without_a_gpu():
    """entry points that are pure host code: the tuning-mode switch, the validation of imported tuning tables (rows naming a tile shape
    or kernel generation this build does not have are skipped: ADVICE r2), the CU-budget range, table sizes"""
    from road_segmentation_unet_amd import _lib
    L = _lib.lib()
    assert L.rsu_get_autotune() == _lib.TUNE_LOOKUP            # launches look shapes up, they never measure by default
    assert L.rsu_set_autotune(3) == -22 and L.rsu_set_autotune(-1) == -22
    assert L.rsu_set_autotune(_lib.TUNE_MEASURE) == 0 and L.rsu_get_autotune() == _lib.TUNE_MEASURE
    assert L.rsu_set_autotune(_lib.TUNE_LOOKUP) == 0
    n0 = L.rsu_autotune_entries()
    key = list(range(100, 116))
    rows = (ctypes.c_int * (17 * 5))(*(key + [0] + key[:-1] + [7] + [1 | (1 << 8)]          # shape 0 (igemm_fwd2); shape 1 on igemm_pp
                                       + key[:-1] + [8] + [99]                                # no such shape
                                       + key[:-1] + [9] + [0 | (5 << 8)]                      # no such kernel generation
                                       + key[:-1] + [10] + [6 | (1 << 8)]))                   # igemm_pp is not built for shape 6
    assert L.rsu_autotune_import(rows, 5) == 2
    assert L.rsu_autotune_entries() == n0 + 2
    assert L.rsu_set_cu_budget(16) == -22 and L.rsu_set_cu_budget(300) == -22
    assert L.rsu_get_cu_budget() == 256
    assert L.rsu_wgrad_group_table_bytes() > 0 and L.rsu_wgrad_group_ws_floats() > 0 and L.rsu_update_table_entry_bytes() > 0
    host = ctypes.create_string_buffer(L.rsu_update_table_entry_bytes() * 2)
    assert L.rsu_update_table_add_plain(host, 0, 16, 32, 48, 100) == 1          # (pointers are only recorded: 16-byte aligned dummies)
    assert L.rsu_update_table_add_plain(host, 1, 17, 32, 48, 100) == -22        # misaligned
    nb = ctypes.c_int(0)
    assert L.rsu_update_table_finish(host, 1, ctypes.byref(nb)) == 0 and nb.value == 1


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


def test_integration_md_counts_the_entry_points():
    """INTEGRATION.md quotes the number of extern "C" entry points of include/rsu.h: the two must agree (it said 62 for 66)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rsu.h")).read()
    n = len(re.findall(r"^(?:int|size_t|const char\*) rsu_\w+\(", header, flags=re.M))
    m = re.search(r"\((\d+) `extern \"C\"` functions", open(os.path.join(root, "INTEGRATION.md")).read())
    assert m and int(m.group(1)) == n, (m and m.group(1), n)
