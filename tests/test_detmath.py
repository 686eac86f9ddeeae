"""include/hope_detmath.h against glibc: the deterministic double-evaluated functions must stay
within one float ulp of the libm calls the reference makes (atan2f/cosf/sinf in pcl::computeRoots,
asinf/acosf in Utilities::normalAnalysis)."""
import ctypes as C

import numpy as np


def _run(oracle_lib, which, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    det = np.empty_like(a)
    ref = np.empty_like(a)
    f = oracle_lib.dll.hope_oracle_detmath
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(which, a.size, a.ctypes.data, b.ctypes.data, det.ctypes.data, ref.ctypes.data)
    return det, ref


def _ulps(x, y):
    return np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64))


def test_atan2_within_one_ulp(oracle_lib):
    rs = np.random.RandomState(0)
    y = np.abs(rs.standard_normal(400000)).astype(np.float32) * np.float32(10) ** rs.randint(-6, 3, 400000)
    x = rs.standard_normal(400000).astype(np.float32) * np.float32(10) ** rs.randint(-6, 3, 400000)
    y, x = y.astype(np.float32), x.astype(np.float32)
    det, ref = _run(oracle_lib, 0, y, x)
    assert _ulps(det, ref).max() <= 1
    # glibc's atan2f is itself off by one ulp in ~16 % of these cases; the deterministic version
    # is the correctly rounded value (double libm result rounded once to float)
    true = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert (_ulps(det, true) > 0).mean() < 1e-5


def test_cos_sin_within_one_ulp(oracle_lib):
    x = np.linspace(0.0, np.pi / 2, 300001).astype(np.float32)
    for which, fn in ((1, np.cos), (2, np.sin)):
        det, ref = _run(oracle_lib, which, x)
        assert _ulps(det[1:-1], ref[1:-1]).max() <= 1
        true = fn(x.astype(np.float64)).astype(np.float32)
        assert (_ulps(det[1:-1], true[1:-1]) > 0).mean() < 1e-5


def test_asin_acos_within_one_ulp(oracle_lib):
    x = np.concatenate([np.linspace(-1, 1, 200001), [1.0, -1.0, 0.0]]).astype(np.float32)
    for which, fn in ((3, np.arcsin), (4, np.arccos)):
        det, ref = _run(oracle_lib, which, x)
        ok = np.isfinite(ref) & (np.abs(ref) > 1e-30)
        assert _ulps(det[ok], ref[ok]).max() <= 1
        true = fn(x.astype(np.float64)).astype(np.float32)
        assert (_ulps(det[ok], true[ok]) > 0).mean() < 1e-5


def test_out_of_domain_is_nan(oracle_lib):
    x = np.array([1.0000001, -1.5, np.nan], np.float32)
    for which in (3, 4):
        det, ref = _run(oracle_lib, which, x)
        assert np.isnan(det).all() and np.isnan(ref).all()
