"""The order-preserving float sum behind the wide TSVQ kernels (csrc/fsum_scan.h), host build, against the plain sequential binary32 sum it
must reproduce BIT FOR BIT (the reference's running sums: encoder/basisu_enc.h:1708-1735 prepare_root, :1810-1823 covariance,
:1862-1885 prep_split, :1985-2005 refine_split). Inputs are shaped like the chains of both codebook builds plus adversarial ones."""
import numpy as np
import pytest

from helpers import fsum_host, ptr, f32p, u64p


def _both(a, start=0.0, block=256):
    """the sequential sum, and the blocked form built BOTH ways (saturating reference form / the kernels' branch-light form): all three must agree"""
    a = np.ascontiguousarray(a, np.float32)
    L = fsum_host()
    stats = np.zeros(3, np.uint64)
    seq = np.float32(L.fsum_sequential(ptr(a, f32p), a.size, start))
    L.fsum_set_fast(0)
    blk = np.float32(L.fsum_blocked(ptr(a, f32p), a.size, start, block, ptr(stats, u64p)))
    L.fsum_set_fast(1)
    stats2 = np.zeros(3, np.uint64)
    fast = np.float32(L.fsum_blocked(ptr(a, f32p), a.size, start, block, ptr(stats2, u64p)))
    if a.size and float(a.min()) >= 0.0 and start >= 0.0:   # monotone chain: the walk's two-integer rule must agree as well
        L.fsum_set_fast(2)
        mono = np.float32(L.fsum_blocked(ptr(a, f32p), a.size, start, block, None))
        assert _same(blk, mono), ("monotone rule differs", blk, mono)
    L.fsum_set_fast(0)
    assert _same(blk, fast), ("kernel form differs", blk, fast)
    if a.size >= 70000 and block <= 512:   # the kernels' form must not lose (many) stretches to its bad flag
        assert stats2[0] >= 0.97 * stats[0], (stats, stats2)
    return seq, blk, stats


def _same(x, y):
    return np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32) or (np.isnan(x) and np.isnan(y))


def _selector_chain(rng, n, wmax):
    """addends of a selector-side chain: fl(v * (float)weight), v in 0..3, integer weights (frontend.cpp:2176-2186: clamp(dist/300, 1, 4096) summed over duplicates)"""
    v = rng.integers(0, 4, n).astype(np.float32)
    w = np.minimum(rng.geometric(1.0 / wmax, n), 4096 * 64).astype(np.uint64).astype(np.float32)
    side = rng.integers(0, 2, n).astype(np.float32)   # the other child's members add +0
    return v * w * side


@pytest.mark.parametrize("n,wmax,block", [(1000, 5, 64), (70000, 40, 256), (700000, 300, 256), (700000, 3000, 1024), (300000, 1, 256), (4100000, 100, 256)])
def test_selector_like_chains(n, wmax, block):
    rng = np.random.default_rng(n + wmax)
    a = _selector_chain(rng, n, wmax)
    seq, blk, stats = _both(a, 0.0, block)
    assert _same(seq, blk), (seq, blk, stats)
    if n >= 70000:  # nearly every block must have gone through a stretch, or the device path would be no faster than the chain
        assert stats[0] > 0.9 * (stats[0] + stats[1]), stats


@pytest.mark.parametrize("seed", range(6))
def test_endpoint_like_chains(seed):
    """vec6F components are colour / 255 (frontend.cpp:843-857) times a float weight: non-integer positive addends"""
    rng = np.random.default_rng(seed)
    n = 40000
    a = (rng.integers(0, 32, n).astype(np.float32) * np.float32(8.0 / 255.0)) * rng.integers(1, 600, n).astype(np.float32)
    seq, blk, stats = _both(a, 0.0, 128)
    assert _same(seq, blk), (seq, blk, stats)


@pytest.mark.parametrize("seed", range(8))
def test_signed_chains(seed):
    """covariance entries (enc.h:1819): products of centred values, drifting or hovering around zero"""
    rng = np.random.default_rng(100 + seed)
    n = 200000
    drift = [0.0, 0.3, -0.7, 0.02][seed % 4]
    x = (rng.integers(0, 4, n) - 1.37).astype(np.float32)
    y = (rng.integers(0, 4, n) - 1.61 + drift * x).astype(np.float32)
    a = x * (rng.integers(1, 900, n).astype(np.float32) * y)
    seq, blk, stats = _both(a, 0.0, 64)
    assert _same(seq, blk), (seq, blk, stats)


def test_adversarial():
    rng = np.random.default_rng(7)
    cases = {
        "all ties": np.full(100000, 1.0, np.float32),                    # 2^24 reached: every further add is a tie
        "ties from odd": np.concatenate([[16777217.0 * 2], np.full(50000, 1.0)]).astype(np.float32),
        "halves": np.full(300000, 0.5, np.float32),
        "tiny after big": np.concatenate([[3.0e9], rng.random(100000) * 100]).astype(np.float32),
        "zeros": np.zeros(5000, np.float32),
        "negative zeros": np.full(5000, -0.0, np.float32),
        "cancel to zero": np.concatenate([np.full(4000, 3.25), np.full(4000, -3.25), np.full(100, 1e-3)]).astype(np.float32),
        "alternating": (np.arange(100000) % 2 * 2 - 1).astype(np.float32) * np.float32(1e6) + rng.random(100000).astype(np.float32),
        "denormals": (rng.random(20000) * 1e-41).astype(np.float32),
        "denormal to normal": np.full(30000, 1e-39, np.float32),
        "wide range": np.float32(2.0) ** rng.integers(-60, 60, 50000).astype(np.float32) * rng.choice([-1, 1], 50000).astype(np.float32),
        "overflow": np.full(1000, 3e38, np.float32),
        "inf": np.concatenate([rng.random(1000), [np.inf], rng.random(1000)]).astype(np.float32),
        "nan": np.concatenate([rng.random(1000), [np.nan], rng.random(1000)]).astype(np.float32),
        "power of two states": np.concatenate([[8388608.0], np.full(3000, 0.25), [8388608.0], np.full(3000, -0.25)]).astype(np.float32),
        "near max finite": np.full(300, 1.7e38, np.float32) * np.float32(0.01),
    }
    with np.errstate(all="ignore"):
        for name, a in cases.items():
            for block in (1, 7, 64, 256):
                for start in (0.0, 1.0, -12345.678, 16777216.0):
                    seq, blk, _ = _both(a, start, block)
                    assert _same(seq, blk), (name, block, start, seq, blk)


def test_random_prefix_lengths_and_starts():
    rng = np.random.default_rng(11)
    a = _selector_chain(rng, 300000, 200)
    for _ in range(40):
        i0, n = int(rng.integers(0, 200000)), int(rng.integers(1, 90000))
        start = float(np.float32(rng.random() * 10.0 ** rng.integers(0, 9)))
        seq, blk, _ = _both(a[i0:i0 + n], start, int(rng.choice([32, 64, 256, 1000])))
        assert _same(seq, blk)


def test_composition_is_associative():
    """the per-block stretch of a range composed from pieces = the stretch pushed addend by addend (what the cross-lane / cross-block
    scans rely on), for positive and negative states"""
    rng = np.random.default_rng(5)
    L = fsum_host()
    for trial in range(200):
        n = int(rng.integers(1, 3000))
        kind = trial % 3
        if kind == 0:
            a = _selector_chain(rng, n, int(rng.integers(1, 2000)))
        elif kind == 1:
            a = ((rng.random(n) - 0.5) * 10.0 ** rng.integers(-3, 6)).astype(np.float32)
        else:
            a = (rng.integers(-8, 9, n) * 2.0 ** rng.integers(-4, 20)).astype(np.float32)   # many exact ties
        a = np.ascontiguousarray(a, np.float32)
        E = int(rng.integers(127, 175))
        for neg in (0, 1):
            assert L.fsum_compose_check(ptr(a, f32p), n, E, neg, int(rng.integers(1, 200))) == 1, (trial, n, E, neg)
