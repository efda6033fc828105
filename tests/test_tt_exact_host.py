"""The reference's DOUBLE accumulators over float addends that are not integers (l_ttsum / r_ttsum += w * |v|^2 of the 6-float endpoint vectors,
encoder/basisu_enc.h:1996-2006) the way the many-workgroup endpoint split walks them (csrc/tt_exact.h, used by tsvq_wide6_kernels.hip: blocks that pass the exactness
test in one step, the rest member by member), host build, against the plain sequential sum s <- s + (double)a[i] it must reproduce BIT FOR BIT -- and the test's
promise itself: a block it passes sums to the same double in any order."""
import numpy as np
import pytest

from helpers import tt_exact_host, ptr, f32p, u64p


def _both(a, block=256):
    a = np.ascontiguousarray(a, np.float32)
    L = tt_exact_host()
    stats = np.zeros(3, np.uint64)
    seq = np.float64(L.tt_sequential(ptr(a, f32p), a.size))
    blk = np.float64(L.tt_blocked(ptr(a, f32p), a.size, block, ptr(stats, u64p)))
    return seq, blk, stats


def _same(x, y):
    return np.float64(x).view(np.uint64) == np.float64(y).view(np.uint64) or (np.isnan(x) and np.isnan(y))


def _expand5(c5):
    return ((c5 << 3) | (c5 >> 2)).astype(np.float32) * np.float32(1.0 / 255.0)


def _tt(v, w):
    """w * |v|^2 with the float operations of the reference (dot product in component order, then the product)"""
    d = (v[:, 0] * v[:, 0]).astype(np.float32)
    for k in range(1, v.shape[1]):
        d = (d + (v[:, k] * v[:, k]).astype(np.float32)).astype(np.float32)
    return (w.astype(np.float32) * d).astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_endpoint_like_chains(seed):
    rng = np.random.default_rng(seed)
    n = 60000
    lo = rng.integers(0, 32, (n, 3)); hi = np.minimum(31, lo + rng.integers(0, 12, (n, 3)))
    v = np.concatenate([_expand5(lo), _expand5(hi)], axis=1)
    w = rng.integers(1, 5000, n)
    a = _tt(v, w)
    seq, blk, stats = _both(a)
    assert _same(seq, blk) and stats[2] == 0
    assert stats[0] >= 0.9 * (stats[0] + stats[1]), stats   # bright vectors: nearly every block in one step


@pytest.mark.parametrize("seed,wmax", [(0, 1), (1, 1 << 20), (2, 1 << 34), (3, 1 << 44)])
def test_dark_vectors_under_large_sums(seed, wmax):
    """near-black vectors (addends around 2^-10 .. 2^-5) sprinkled among bright ones with enormous weights: adds that DO round in the reference, in many blocks"""
    rng = np.random.default_rng(100 + seed)
    n = 80000
    lo = rng.integers(0, 3, (n, 3)); hi = np.minimum(31, lo + rng.integers(0, 3, (n, 3)))
    bright = rng.random(n) < 0.1
    lo[bright] = rng.integers(8, 32, (int(bright.sum()), 3)); hi[bright] = np.minimum(31, lo[bright] + rng.integers(0, 8, (int(bright.sum()), 3)))
    v = np.concatenate([_expand5(lo), _expand5(hi)], axis=1)
    w = rng.integers(1, wmax + 1, n)
    w[rng.random(n) < 0.7] = 1
    a = _tt(v, w)
    for block in (256, 64, 1000):
        seq, blk, stats = _both(a, block)
        assert _same(seq, blk) and stats[2] == 0, (block, stats)
    if wmax >= 1 << 20:
        assert stats[1] > 0   # the case is only worth something if some blocks did have to be walked


def test_boundaries_and_degenerate_inputs():
    rng = np.random.default_rng(7)
    cases = {
        "zeros": np.zeros(5000, np.float32),
        "ones_to_2^24_and_beyond": np.ones(70000, np.float32),
        "powers_of_two": np.float32(2.0) ** rng.integers(-40, 40, 30000).astype(np.float32),
        "tiny_then_huge": np.concatenate([np.full(3000, 2.0 ** -30, np.float32), np.full(3000, 2.0 ** 30, np.float32), np.full(3000, 2.0 ** -30, np.float32)]),
        "huge_then_tiny_ties": np.concatenate([np.full(10, 2.0 ** 40, np.float32), np.full(5000, 2.0 ** -13, np.float32)]),   # halves of an ulp: ties to even
        "denormals": np.concatenate([np.full(2000, 1e-40, np.float32), rng.random(2000).astype(np.float32)]),
        "negative_mixed_in": np.concatenate([rng.random(3000).astype(np.float32), -rng.random(300).astype(np.float32), rng.random(3000).astype(np.float32)]),
        "mantissa_patterns": (rng.integers(1, 1 << 24, 40000).astype(np.float32) * np.float32(2.0 ** -20)),
        "one": np.array([0.3], np.float32),
        "empty": np.zeros(0, np.float32),
    }
    for name, a in cases.items():
        for block in (256, 7):
            seq, blk, stats = _both(a, block)
            assert _same(seq, blk) and stats[2] == 0, (name, block, seq, blk, stats)


def test_low_bit():
    L = tt_exact_host()
    assert L.tt_low_bit(1.0) == 0 and L.tt_low_bit(3.0) == 0 and L.tt_low_bit(6.0) == 1 and L.tt_low_bit(0.375) == -3
    assert L.tt_low_bit(float(2.0 ** 52 + 1)) == 0 and L.tt_low_bit(float(2.0 ** 53)) == 53 and L.tt_low_bit(1.0 + 2.0 ** -52) == -52
    assert L.tt_low_bit(0.0) == 1 << 20
