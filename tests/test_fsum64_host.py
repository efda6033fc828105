"""The order-preserving DOUBLE sum of float addends (csrc/fsum_scan64.h: the reference's double accumulators ttsum / l_weight / r_weight,
encoder/basisu_enc.h:1873-1881, 1996-2006, for vectors whose addends are not integers), host build, against the plain sequential sum
s <- s + (double)a[i] it must reproduce BIT FOR BIT. No kernel uses the header yet; this pins its arithmetic for the endpoint-side
many-workgroup split."""
import numpy as np
import pytest

from helpers import fsum64_host, ptr, f32p, u64p


def _both(a, start=0.0, block=256):
    a = np.ascontiguousarray(a, np.float32)
    L = fsum64_host()
    stats = np.zeros(3, np.uint64)
    seq = np.float64(L.fsum64_sequential(ptr(a, f32p), a.size, start))
    blk = np.float64(L.fsum64_blocked(ptr(a, f32p), a.size, start, block, ptr(stats, u64p)))
    return seq, blk, stats


def _same(x, y):
    return np.float64(x).view(np.uint64) == np.float64(y).view(np.uint64) or (np.isnan(x) and np.isnan(y))


@pytest.mark.parametrize("seed", range(6))
def test_ttsum_like_chains(seed):
    """ttsum of the endpoint side: weight * |v|^2 rounded to float, v = colour / 255 in six components, integer weights"""
    rng = np.random.default_rng(seed)
    n = 60000
    v = rng.integers(0, 256, (n, 6)).astype(np.float32) * np.float32(1.0 / 255.0)
    w = rng.integers(1, 5000, n).astype(np.float32)
    a = (w * (v * v).sum(axis=1, dtype=np.float32)).astype(np.float32)
    seq, blk, stats = _both(a, 0.0, 256)
    assert _same(seq, blk), (seq, blk, stats)
    assert stats[0] > 0.9 * (stats[0] + stats[1]), stats   # nearly every block through its map


@pytest.mark.parametrize("n,scale,block", [(1000, 1.0, 64), (200000, 1e-6, 256), (200000, 1e6, 256), (500000, 1.0, 1024)])
def test_positive_floats_of_any_size(n, scale, block):
    rng = np.random.default_rng(n)
    a = (rng.random(n, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    seq, blk, stats = _both(a, 0.0, block)
    assert _same(seq, blk), (seq, blk, stats)


@pytest.mark.parametrize("seed", range(4))
def test_wide_dynamic_range_rounds(seed):
    """addends spread over 60 binades: the small ones are below half an ulp of the sum, the middle ones hit every rounding class, ties included"""
    rng = np.random.default_rng(100 + seed)
    n = 100000
    a = np.ldexp(rng.integers(1 << 23, 1 << 24, n).astype(np.float32), rng.integers(-60, 0, n)).astype(np.float32)
    a[rng.integers(0, n, 2000)] = np.float32(2.0 ** -30)   # exact powers of two: ties once the sum's ulp is 2^-29
    seq, blk, stats = _both(a, 1.0, 256)
    assert _same(seq, blk), (seq, blk, stats)


@pytest.mark.parametrize("seed", range(4))
def test_signed_sums_with_cancellation(seed):
    rng = np.random.default_rng(200 + seed)
    n = 80000
    a = (rng.normal(0, 1, n) * np.exp(rng.normal(0, 3, n))).astype(np.float32)
    seq, blk, stats = _both(a, 0.0, 128)
    assert _same(seq, blk), (seq, blk, stats)
    seq, blk, stats = _both(a, -3.5e7, 128)
    assert _same(seq, blk), (seq, blk, stats)


def test_special_values_fall_back_to_plain_adds():
    a = np.array([1.0, np.inf, 2.0, -np.inf, 3.0], np.float32)
    seq, blk, _ = _both(a, 0.0, 2)
    assert _same(seq, blk)
    a = np.array([1e38, 1e38, 1e-45, 0.0, -0.0, 5e-39], np.float32)
    seq, blk, _ = _both(a, 0.0, 2)
    assert _same(seq, blk)
    seq, blk, _ = _both(np.zeros(0, np.float32), 7.25, 4)
    assert _same(seq, blk) and seq == 7.25


@pytest.mark.parametrize("seed", range(5))
def test_composition_is_associative(seed):
    rng = np.random.default_rng(300 + seed)
    n = 4096
    a = (rng.normal(0, 1, n) * np.exp(rng.normal(0, 2, n))).astype(np.float32)
    L = fsum64_host()
    for E in (1023, 1023 + 10, 1023 - 5):
        for neg in (0, 1):
            for piece in (1, 7, 64, 1000):
                assert L.fsum64_compose_check(ptr(a, f32p), n, E, neg, piece) == 1, (E, neg, piece)
