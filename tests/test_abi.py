"""The C-ABI library loads and exports every symbol include/hope_gpu.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hope_gpu.h")).read()
    return re.findall(r"HOPE_API\s+[\w\s\*]+?\b(hope_\w+)\s*\(", txt)


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    assert len(syms) == len(set(syms)) >= 24
    from hope_amd.capi import ABI_SYMBOLS
    for s in ABI_SYMBOLS:
        assert "hope_" + s in syms


def test_hip_library_exports_every_declared_symbol():
    from hope_amd.capi import HIP_LIB_PATH
    assert os.path.exists(HIP_LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = C.CDLL(HIP_LIB_PATH)
    for s in declared_symbols():
        assert hasattr(dll, s), s


def test_oracle_exports_the_same_abi(oracle_lib):
    from hope_amd.capi import ABI_SYMBOLS
    for s in ABI_SYMBOLS:
        assert hasattr(oracle_lib.dll, "hope_oracle_" + s), s


def test_binding_struct_sizes_match_header(oracle_lib):
    from hope_amd.capi import HopeCluster, HopeConfig, HopeCounts, HopeMaxPlane
    assert C.sizeof(HopeConfig) == 4 * 9 + 4 * 3 + 4 * 2 + 4 * 2 + 16
    assert C.sizeof(HopeCounts) == 8 * 4 + 4 * 3 + 4 * 6 + 4 * 2 + 4 * 3
    assert C.sizeof(HopeCluster) == 32
    assert C.sizeof(HopeMaxPlane) == 16 + 64


def test_missing_library_fails_loudly(tmp_path):
    from hope_amd.capi import HopeError, HopeLib
    with pytest.raises(HopeError):
        HopeLib(str(tmp_path / "nope.so"))


def test_bad_arguments_are_rejected(oracle_lib):
    from hope_amd.capi import HopeConfig
    cfg = oracle_lib.default_config()
    assert abs(cfg.th_xy - 0.05) < 1e-9 and abs(cfg.th_z - 0.02) < 1e-9 and cfg.knn == 8 and cfg.min_cluster == 3
    bad = HopeConfig.from_buffer_copy(cfg)
    bad.th_xy = 0.0
    h = C.c_void_p()
    assert oracle_lib.fn["create"](C.byref(bad), C.byref(h)) < 0
