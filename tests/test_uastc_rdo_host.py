"""CPU-only checks of the UASTC RDO post-pass (uastc_rdo, encoder/basisu_uastc_enc.cpp:3824-4163).

The GPU strips kernel and tests/native/uastc_host.cpp compile the SAME per-block pieces (basis_universal_amd/csrc/uastc_rdo.h); here the
host build, driven by a plain scalar strip loop, is held against (1) the committed known answers of the real reference
(tests/golden/uastc_rdo_vectors.npz, tools/gen_golden_uastc_rdo.py) and (2), where oracle/_ref is present, the reference itself on
fresh inputs and parameter draws. The GPU build is held to the same vectors in test_gpu_uastc_rdo.py. All comparisons are bit-exact.
"""
import ctypes as C
import pathlib

import numpy as np
import pytest

import helpers
from helpers import ptr

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden" / "uastc_rdo_vectors.npz"


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name,flags,jobs,kw", helpers.uastc_rdo_cases(), ids=[c[0] for c in helpers.uastc_rdo_cases()])
@pytest.mark.parametrize("table", [False, True], ids=["decode", "table"])
def test_host_rdo_matches_reference_vectors(golden, name, flags, jobs, kw, table):
    """table: trials scored from the per-block weight-error table (uastc_rdo.h, what the GPU strips kernel does) instead of decoded."""
    packed = golden[f"packed_l{flags & 7}"]
    got = helpers.host_uastc_rdo(packed, golden["blocks"], flags, jobs, table, **kw)
    bad = np.nonzero((got != golden[name]).any(1))[0]
    assert bad.size == 0, f"{name}: {bad.size} blocks differ, first {bad[:5]}"
    assert (golden[name] != packed).any(), "the case must modify something"


def test_unpack_is_the_inverse_of_pack(golden):
    """unpack_block (transcoder.cpp:15274-15735) followed by the hint recomputation and pack_block (uastc_recompute_hints, uastc_enc.cpp:
    3647-3726) must give back an encoder output bit for bit: same endpoints, weights, pattern -- and the same hints, since they are a
    function of those. Invalid mode codes must be refused like the reference does."""
    H = helpers.uastc_host()
    for level in (0, 2, 3):
        packed = golden[f"packed_l{level}"]
        again = np.ascontiguousarray(packed).copy()
        assert H.hc_rehint(ptr(np.ascontiguousarray(golden["blocks"])), packed.shape[0], level, ptr(again)) == 1
        bad = np.nonzero((again != packed).any(1))[0]
        assert bad.size == 0, f"level {level}: {bad.size} blocks change, first {bad[:5]}"
    out = np.zeros(64, np.uint8)
    junk = np.zeros(16, np.uint8)
    junk[0] = 0x45  # 7-bit prefix that is no mode code (transcoder.cpp:14376-14402)
    assert H.hc_unpack_block(ptr(junk), ptr(out)) == 0
    assert H.hc_unpack_block(ptr(np.ascontiguousarray(golden["packed_l2"][0])), ptr(out)) == 1


def test_strips_are_independent(golden):
    """uastc_rdo's total_jobs strips (uastc_enc.cpp:4103-4150) never look across their borders: strip k of a 4-job run equals a single-strip
    run over that sub-array."""
    packed, blocks = golden["packed_l2"], golden["blocks"]
    n = packed.shape[0]
    per = n // 4
    whole = helpers.host_uastc_rdo(packed, blocks, 2, 4, lam=3.0)
    for f in range(0, n, per):
        part = helpers.host_uastc_rdo(packed[f:f + per], blocks[f:f + per], 2, 0, lam=3.0)
        assert (whole[f:f + per] == part).all()


@pytest.mark.ref
@pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref not built")
def test_host_rdo_vs_reference_random_params():
    rng = np.random.default_rng(5)
    img = helpers.synth(192, 128, 21)
    blocks = helpers.to_pixel_blocks(img).copy()
    blocks[700:1100, :, :, 3] = rng.integers(0, 256, size=(400, 4, 4), dtype=np.uint8)
    blocks[1100:, :, :, 3] = blocks[1100:, :, :, 1]
    # every 5th block luminance+alpha (modes 15/17, whose selector fields reach into the endpoint bits of their neighbours: the deferred
    # refit of uastc_rdo.h has to settle those first)
    grey = blocks[::5, :, :, 1].copy()
    blocks[::5, :, :, 0] = grey
    blocks[::5, :, :, 2] = grey
    blocks[::5, :, :, 3] = 255 - grey // 2
    for trial in range(7):
        if trial == 6:
            blocks = helpers.smooth_with_la_blocks(192, 128, 3)
        level = int(rng.integers(0, 3)) if trial < 6 else 0
        packed = helpers.ref_encode_uastc(blocks, level)
        kw = dict(lam=float(rng.choice([0.25, 0.75, 2.0, 6.0])), dict_size=int(rng.choice([16, 256, 4096, 16384])), refine=int(rng.integers(0, 2)),
                  skip_rms=float(rng.choice([4.0, 8.0, 20.0])), max_rms_ratio=float(rng.choice([1.1, 3.0, 10.0])),
                  smooth_scale=float(rng.choice([1.0, 10.0, 20.0])), smooth_std_dev=float(rng.choice([9.0, 18.0])), literal_cost=int(rng.choice([80, 100, 130])))
        jobs = int(rng.choice([0, 2, 4, 7]))
        if trial == 6:
            kw, jobs = dict(lam=20.0), 0
        want = helpers.ref_uastc_rdo(packed, blocks, level, jobs, **kw)
        for table in (False, True):
            got = helpers.host_uastc_rdo(packed, blocks, level, jobs, table, **kw)
            bad = np.nonzero((got != want).any(1))[0]
            assert bad.size == 0, f"trial {trial} level {level} jobs {jobs} table {table} {kw}: {bad.size} blocks differ, first {bad[:5]}"
