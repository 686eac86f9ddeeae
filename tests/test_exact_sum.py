"""exact_sum.hpp: the reference's sequential fp32 sums (getCloudZInfo utilities.cpp:751-769, calNormalMean
:257-282, normalAnalysis :151-255) evaluated in parallel as integer steps per binade must equal the chain of
float adds bit for bit -- on friendly inputs and on inputs built to break it (exact halves, cancellation,
binade boundaries, sign changes, denormals, infinities, NaN).  The checker is numpy's float32 cumulative
sum, which adds one element after the other."""
import numpy as np
import pytest

from hope_amd.capi import HopeContext

pytestmark = pytest.mark.gpu


def seq_sum(v):
    with np.errstate(all="ignore"):
        # the reference's accumulators start at +0.0f (utilities.cpp:753,260)
        return np.cumsum(np.concatenate([[np.float32(0)], v.astype(np.float32)]), dtype=np.float32)[-1]


def same_bits(a, b):
    a, b = np.float32(a), np.float32(b)
    if np.isnan(a) and np.isnan(b):
        return True
    return a.view(np.uint32) == b.view(np.uint32)


def check(ctx, rows, what):
    rows = [np.asarray(r, np.float32) for r in rows]
    n = max(len(r) for r in rows)
    m = np.zeros((len(rows), n), np.float32)
    for i, r in enumerate(rows):
        m[i, :len(r)] = r   # trailing +0.0 leaves a sum as it is
    sums, stats = ctx.chain_sums(m)
    for i, r in enumerate(rows):
        want = seq_sum(m[i])
        assert same_bits(sums[i], want), (what, i, float(sums[i]), float(want), stats)
    return stats


def test_sums_of_cluster_like_terms(hip_lib):
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(11)
    n = 600000
    nz = rng.uniform(0.87, 1.0, n)
    nx = rng.normal(0.0, 0.05, n)
    ny = rng.normal(0.01, 0.05, n)
    z = rng.normal(0.0, 0.002, n)
    side = rng.random(n) < 0.5
    stats = check(g, [z, nx, ny, nz, nx * nx, nx * ny, ny * ny, np.where(side, nz, 0), np.where(side, 0, nx),
                      -nz, 1000.0 + z], "cluster terms")
    # nearly everything goes through the 1024-member records
    assert stats["top_hits"] > 0.8 * 11 * n / 1024, stats
    assert stats["slow_blocks"] < 0.08 * 11 * n / 64, stats


def test_lengths_and_block_boundaries(hip_lib):
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(12)
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 4096, 65535, 65536, 65537):
        check(g, [rng.uniform(0.5, 1.0, n), rng.normal(0, 1, n), np.full(n, 0.1)], f"n={n}")


def test_exact_halves_everywhere(hip_lib):
    """Integers and half-integers: once the sum passes 2^24 every other term sits exactly between two
    representable sums, and round-to-even looks at the parity of the running sum."""
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(13)
    n = 400000
    ints = rng.integers(1, 200, n).astype(np.float32)
    halves = rng.integers(1, 64, n).astype(np.float32) * 0.5
    mixed = np.where(rng.random(n) < 0.5, ints, -halves)
    ones = np.ones(n * 50, np.float32)  # reaches 2^24 = 16777216 and stays there: s + 1 ties to even
    threes = np.full(n * 20, 3.0, np.float32)
    stats = check(g, [ints, halves, mixed, ones, threes, ints[::-1] * 0.25], "halves")
    assert stats["top_hits"] > 0


def test_cancellation_and_sign_changes(hip_lib):
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(14)
    n = 200000
    alt = np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * rng.uniform(0.9, 1.1, n)
    big_small = np.where(np.arange(n) % 2 == 0, 1e6, -1e6) + rng.normal(0, 1, n)
    walk = rng.normal(0, 1, n)
    drift_down = np.concatenate([np.full(1000, 1000.0), rng.uniform(-2.1, -1.9, n)])  # crosses zero, goes negative
    sawtooth = np.tile(np.concatenate([np.full(500, 0.37), np.full(500, -0.37)]), 200)
    stats = check(g, [alt, big_small, walk, drift_down, sawtooth], "cancellation")
    assert stats["plain_blocks"] > 0   # the plain chain took over where integer steps accept nothing


def test_wide_dynamic_range_and_special_values(hip_lib):
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(15)
    n = 100000
    lognorm = np.exp(rng.normal(0, 6, n)) * np.where(rng.random(n) < 0.3, -1, 1)
    tiny = np.full(n, 1e-41, np.float32)              # denormal terms, denormal sum
    tiny_then_big = np.concatenate([np.full(5000, 1e-41), np.full(5000, 1e-3), np.full(5000, 3.0e38 / 6000)])
    with_inf = rng.uniform(0, 1, n).astype(np.float32)
    with_inf[n // 2] = np.inf
    with_nan = rng.uniform(0, 1, n).astype(np.float32)
    with_nan[n // 3] = np.nan
    inf_minus_inf = rng.uniform(0, 1, n).astype(np.float32)
    inf_minus_inf[100] = np.inf
    inf_minus_inf[50000] = -np.inf
    overflow = np.full(n, 3.0e38 / 1000, np.float32)  # runs into +inf
    neg_zero = np.where(np.arange(n) % 3 == 0, -0.0, rng.uniform(0, 1, n))
    check(g, [lognorm, tiny, tiny_then_big, with_inf, with_nan, inf_minus_inf, overflow, neg_zero,
              np.zeros(n), np.full(n, -0.0)], "special values")


def test_binade_boundaries(hip_lib):
    """Sums that sit on a power of two, step across it in both directions and land on it exactly."""
    g = HopeContext(hip_lib)
    rng = np.random.default_rng(16)
    rows = []
    for p in (1.0, 1024.0, 2.0 ** 20):
        hover = np.concatenate([[p], np.where(np.arange(50000) % 2 == 0, -1, 1) * rng.uniform(0, p * 2.0 ** -12, 50000)])
        land = np.concatenate([[p - 8 * p * 2.0 ** -20], np.full(16, p * 2.0 ** -20)])
        down = np.concatenate([[2 * p], np.full(70000, -p * 2.0 ** -15)])
        rows += [hover, land, down]
    check(g, rows, "binade boundaries")


def test_random_mixtures(hip_lib):
    g = HopeContext(hip_lib)
    for seed in range(8):
        rng = np.random.default_rng(100 + seed)
        n = int(rng.integers(1000, 300000))
        scale = 10.0 ** rng.uniform(-6, 6)
        bias = rng.uniform(-1, 1)
        a = (rng.normal(bias, 1, n) * scale).astype(np.float32)
        q = (np.round(rng.normal(bias, 1, n) * 64) / 64 * scale).astype(np.float32)  # few mantissa bits: many ties
        mask = rng.random(n) < rng.uniform(0.05, 0.95)
        check(g, [a, q, np.where(mask, a, 0), np.where(mask, 0, q), np.abs(a), -np.abs(q)], f"seed {seed}")
