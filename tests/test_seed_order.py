"""The ZGrowing seed order (z_growing.cpp:235-246): std::sort over all-equal keys.  The HIP
kernel k_seed_ranks derives that permutation in closed form; this checks the closed form (restated
here in Python exactly as the kernel computes it) against a literal std::sort call in the oracle."""
import ctypes as C

import numpy as np


def seed_at(n, p):
    f, l = 0, n
    chain = []
    while l - f > 16:
        chain.append((f, l))
        cut = f + 1 + (l - f - 1) // 2
        if p >= cut:
            f = cut
        else:
            l = cut
    for f, l in reversed(chain):
        mid = f + (l - f) // 2
        s = f if p == f else l + f - p
        if s == f:
            s = mid
        elif s == mid:
            s = f
        p = s
    return p


def std_sort_order(oracle_lib, n):
    out = np.empty(n, np.int32)
    fn = oracle_lib.dll.hope_oracle_seed_order
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p]
    assert fn(n, out.ctypes.data) == 0
    return out


def test_closed_form_matches_std_sort(oracle_lib):
    sizes = list(range(0, 140)) + [255, 256, 257, 511, 1000, 1217, 4095, 4096, 4097, 5699, 13528, 65537]
    for n in sizes:
        ref = std_sort_order(oracle_lib, n)
        mine = np.array([seed_at(n, p) for p in range(n)], np.int32)
        assert np.array_equal(ref, mine), n


def test_is_a_permutation(oracle_lib):
    for n in (17, 100, 3001):
        assert np.array_equal(np.sort(std_sort_order(oracle_lib, n)), np.arange(n))
