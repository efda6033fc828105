"""-m gpu parity tests of the device-resident layer (include/basisu_hip.h section 2) against the CPU oracle.

Every comparison is bit-exact. Inputs are seeded; sizes are chosen so the plain-C oracle finishes in seconds.
The real reference (oracle/_ref) pins the oracle itself in test_oracle_vs_reference.py.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import (oracle, ptr, u8p, u32p, u64p, f32p, synth, uniform_random, to_pixel_blocks, csr_from_lists)

pytestmark = pytest.mark.gpu

VP = C.c_void_p


def _images():
    a = to_pixel_blocks(synth(256, 192, 1234))
    b = to_pixel_blocks(uniform_random(64, 64, 42))
    flat = np.zeros((64, 4, 4, 4), np.uint8)
    flat[:, :, :, 3] = 255
    flat[:32, :, :, :3] = 255
    flat[40:48, :, :, 0] = 17
    grad = np.zeros((32, 64, 4), np.uint8)
    grad[..., 0] = np.arange(64)[None, :] * 4
    grad[..., 1] = np.arange(32)[:, None] * 8
    grad[..., 3] = 255
    return np.concatenate([a, b, flat, to_pixel_blocks(grad)])


@pytest.fixture(scope="module")
def blocks():
    return _images()


@pytest.fixture(scope="module")
def d_blocks(hip_ctx, blocks):
    p = hip_ctx.upload(blocks)
    yield p
    hip_ctx.free(p)


@pytest.mark.parametrize("perceptual", [1, 0])
@pytest.mark.parametrize("quality,level", [(0, 0), (1, 1), (2, 2), (3, 6)])
def test_encode_etc1s_blocks(hip_ctx, blocks, d_blocks, quality, level, perceptual):
    n = blocks.shape[0]
    d_out = hip_ctx.alloc(n * 8)
    hip_ctx.check(hip_ctx.lib.k_encode_etc1s_blocks(hip_ctx.h, d_blocks, n, quality, perceptual, d_out), "k_encode")
    got = hip_ctx.download(d_out, (n, 8), np.uint8)
    hip_ctx.free(d_out)
    exp = np.zeros((n, 8), np.uint8)
    oracle().orc_encode_etc1s_blocks(ptr(blocks), n, level, perceptual, ptr(exp))
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {n} blocks differ, first {bad[:5]}: got {got[bad[:2]]} exp {exp[bad[:2]]}"


def test_encode_ragged_tail(hip_ctx, blocks):
    # n not a multiple of the 32 blocks a workgroup handles, and n == 1
    for n in (1, 33, 95):
        b = np.ascontiguousarray(blocks[:n])
        d_in = hip_ctx.upload(b)
        d_out = hip_ctx.alloc(n * 8 + 64)
        hip_ctx.check(hip_ctx.lib.memset(hip_ctx.h, d_out, 0xAB, n * 8 + 64), "memset")
        hip_ctx.check(hip_ctx.lib.k_encode_etc1s_blocks(hip_ctx.h, d_in, n, 1, 1, d_out), "k_encode")
        got = hip_ctx.download(d_out, (n * 8 + 64,), np.uint8)
        exp = np.zeros((n, 8), np.uint8)
        oracle().orc_encode_etc1s_blocks(ptr(b), n, 1, 1, ptr(exp))
        assert (got[:n * 8].reshape(n, 8) == exp).all()
        assert (got[n * 8:] == 0xAB).all(), "wrote past the end"
        hip_ctx.free(d_in); hip_ctx.free(d_out)


@pytest.mark.parametrize("kind", ["pageable", "pinned"])
def test_upload_and_encode_pipeline(hip_ctx, blocks, kind):
    """bu_hip_k_upload_and_encode_etc1s_blocks: host tiles in 65,536-block pieces on the side stream, piece i's kernel behind piece i's copy. 10 pieces and a ragged one
    (more pieces than ring slots: the helper threads wait for slots), page-locked and pageable source: the tiles arrive intact and the ETC1S blocks are those of one launch
    over resident tiles -- themselves held to the oracle on a sample."""
    n = 10 * 65536 + 777
    rng = np.random.default_rng(5)
    src = np.ascontiguousarray(blocks[rng.integers(0, blocks.shape[0], n)]).reshape(n, 64)
    src[:, 0] ^= (np.arange(n) & 255).astype(np.uint8)          # every piece differs from every other
    pinned = None
    if kind == "pinned":   # page-locked memory from the library (bu_hip_host_alloc)
        pinned = hip_ctx.lib.host_alloc(n * 64)
        assert pinned
        host = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(n * 64,)).reshape(n, 64)
        host[:] = src
    else:
        host = src
    d_px, d_out, d_ref = hip_ctx.alloc(n * 64), hip_ctx.alloc(n * 8), hip_ctx.alloc(n * 8)
    hip_ctx.check(hip_ctx.lib.memset(hip_ctx.h, d_px, 0, n * 64), "memset")
    for rep in range(2):   # the second call finds the ring and its events used
        hip_ctx.check(hip_ctx.lib.k_upload_and_encode_etc1s_blocks(hip_ctx.h, d_px, host.ctypes.data_as(VP), n, 1, 1, d_out), "k_upload_and_encode")
    assert (hip_ctx.download(d_px, (n, 64), np.uint8) == src).all(), "tiles damaged on the way"
    hip_ctx.check(hip_ctx.lib.k_encode_etc1s_blocks(hip_ctx.h, d_px, n, 1, 1, d_ref), "k_encode")
    got, ref = hip_ctx.download(d_out, (n, 8), np.uint8), hip_ctx.download(d_ref, (n, 8), np.uint8)
    assert (got == ref).all(), np.nonzero((got != ref).any(axis=1))[0][:8]
    pick = np.concatenate([np.arange(0, n, 997), np.arange(n - 64, n)])
    sample = np.ascontiguousarray(src[pick])
    exp = np.zeros((pick.size, 8), np.uint8)
    oracle().orc_encode_etc1s_blocks(ptr(sample), pick.size, 1, 1, ptr(exp))
    assert (got[pick] == exp).all()
    for q in (d_px, d_out, d_ref):
        hip_ctx.free(q)
    if pinned:
        del host
        hip_ctx.lib.host_free(pinned)


def test_background_downloads(hip_ctx):
    """bu_hip_download_begin / _wait: a copy on a stream of its own behind everything enqueued so far, carried out by the context's helper thread. Several at once, waited for in
    the opposite order, each seeing the bytes its source held when it was begun (the source is rewritten only after the wait)."""
    L = hip_ctx.lib
    rng = np.random.default_rng(3)
    n = 3 * 1024 * 1024 + 13
    srcs = [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(3)]
    d = [hip_ctx.upload(a) for a in srcs]
    outs = [np.zeros(n, np.uint8) for _ in range(3)]
    for rep in range(2):   # the second round reuses the thread, the stream and the events
        handles = [L.download_begin(hip_ctx.h, outs[i].ctypes.data_as(VP), d[i], n) for i in range(3)]
        assert all(handles)
        for i in (2, 1, 0):
            assert L.download_wait(handles[i]) == 1
            assert (outs[i] == srcs[i]).all()
            outs[i][:] = 0
    assert not L.download_begin(hip_ctx.h, None, d[0], n) and not L.download_begin(hip_ctx.h, outs[0].ctypes.data_as(VP), d[0], 0)   # refused, not crashed
    for q in d:
        hip_ctx.free(q)


def _encode_ref(blocks, level=1, perceptual=1):
    n = blocks.shape[0]
    out = np.zeros((n, 8), np.uint8)
    oracle().orc_encode_etc1s_blocks(ptr(blocks), n, level, perceptual, ptr(out))
    return out


def test_endpoint_training_vectors(hip_ctx, blocks):
    n = blocks.shape[0]
    etc = _encode_ref(blocks)
    d_etc = hip_ctx.upload(etc)
    d_out = hip_ctx.alloc(n * 24)
    hip_ctx.check(hip_ctx.lib.k_endpoint_training_vectors(hip_ctx.h, d_etc, n, d_out), "k_etv")
    got = hip_ctx.download(d_out, (n, 6), np.float32)
    exp = np.zeros((n, 6), np.float32)
    oracle().orc_endpoint_training_vectors(ptr(etc), n, ptr(exp, f32p))
    assert (got.view(np.uint32) == exp.view(np.uint32)).all()
    hip_ctx.free(d_etc); hip_ctx.free(d_out)


def _clusters_by_luma(blocks, k, rng):
    """A plausible endpoint clustering: blocks sorted by mean luma cut into k uneven runs; both subblocks stay together."""
    n = blocks.shape[0]
    order = np.argsort(blocks[..., :3].reshape(n, -1).astype(np.int64).sum(axis=1), kind="stable")
    cuts = np.sort(rng.choice(np.arange(1, n), size=k - 1, replace=False))
    lists, block_cluster = [], np.zeros(n, np.uint32)
    for ci, run in enumerate(np.split(order, cuts)):
        run = rng.permutation(run)
        lists.append(np.stack([run * 2, run * 2 + 1], axis=1).reshape(-1).astype(np.uint32))
        block_cluster[run] = ci
    return lists, block_cluster


# wide_min: bu_hip_tuning::codebook_wide_min -- None = the default (32,768 texels: only the giant cluster below takes the many-workgroup passes of
# etc1s_codebook_wide.inc), 8 = every cluster does, 0 = none does (one workgroup per cluster throughout)
@pytest.mark.parametrize("wide_min", [None, 8, 0])
@pytest.mark.parametrize("perceptual", [1, 0])
@pytest.mark.parametrize("quality,level", [(1, 1), (2, 2), (3, 6)])
def test_generate_endpoint_codebook(hip_ctx, blocks, d_blocks, quality, level, perceptual, wide_min, request):
    if wide_min is not None:
        request.addfinalizer(hip_ctx.set_tuning)
        hip_ctx.set_tuning(codebook_wide_min=wide_min)
    rng = np.random.default_rng(7)
    k = 97
    lists, _ = _clusters_by_luma(blocks, k, rng)
    # one giant cluster as well: exercises the multi-wave reduction with thousands of pixels
    lists.append(np.arange(blocks.shape[0] * 2, dtype=np.uint32))
    k += 1
    offs, idx = csr_from_lists(lists)
    d_offs, d_idx = hip_ctx.upload(offs), hip_ctx.upload(idx)
    params = np.zeros((k, 4), np.uint8); err = np.zeros(k, np.uint64); valid = np.zeros(k, np.uint8)
    d_params, d_err, d_valid = hip_ctx.upload(params), hip_ctx.upload(err), hip_ctx.upload(valid)
    L = hip_ctx.lib

    def run(step):
        hip_ctx.check(L.k_generate_endpoint_codebook(hip_ctx.h, d_blocks, k, offs.ctypes.data_as(VP), d_offs, d_idx, quality, perceptual, step,
                                                     d_params, d_err, d_valid), "k_gec")
        return hip_ctx.download(d_params, (k, 4), np.uint8), hip_ctx.download(d_err, (k,), np.uint64), hip_ctx.download(d_valid, (k,), np.uint8)

    got = run(0)
    oracle().orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), level, perceptual, 0, ptr(params), ptr(err, u64p), ptr(valid))
    assert (got[0] == params).all() and (got[1] == err).all() and (got[2] == valid).all()

    # step 1: perturb half of the previous parameters so that both "keep old" and "take new" branches run (frontend.cpp:1554-1605)
    params2 = params.copy()
    params2[::2, 0] = np.minimum(params2[::2, 0] + 1, 31)
    params2[1::4, 3] = (params2[1::4, 3] + 1) % 8
    valid2 = valid.copy(); valid2[5] = 0
    hip_ctx.check(L.memcpy_h2d(hip_ctx.h, d_params, params2.ctypes.data_as(VP), params2.nbytes), "h2d")
    hip_ctx.check(L.memcpy_h2d(hip_ctx.h, d_valid, valid2.ctypes.data_as(VP), valid2.nbytes), "h2d")
    got = run(1)
    err2 = err.copy()
    oracle().orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), level, perceptual, 1, ptr(params2), ptr(err2, u64p), ptr(valid2))
    assert (got[0] == params2).all() and (got[1] == err2).all() and (got[2] == valid2).all()
    for p in (d_offs, d_idx, d_params, d_err, d_valid):
        hip_ctx.free(p)


@pytest.mark.parametrize("kind", ["bright", "dark_then_bright", "solid"])
def test_generate_endpoint_codebook_of_huge_clusters(hip_ctx, kind, request):
    """Clusters of 10^5-10^6 texels (a sky, a constant alpha plane): their channel sums pass 2^24, where the reference's colour mean -- a RUNNING float sum in texel
    order (etc.cpp:1034-1041, SURVEY hazard H4) -- depends on the order of the adds. The many-workgroup path evaluates that sum without its order
    (etc1s_codebook_wide.inc, cbw_ordered_sum: parity maps per binade); the one-workgroup kernel replays it texel by texel; the oracle is the reference's loop.
    All three must agree, for step 0 and for the keep-unless-better rule of step 1."""
    rng = np.random.default_rng(len(kind))
    n = 1024 * 1024 // 16
    if kind == "bright":      # sums of ~2 x 10^8: four binade changes
        px = rng.integers(120, 256, (n, 4, 4, 4), dtype=np.uint8)
    elif kind == "dark_then_bright":   # a long exact prefix (the sum stays below 2^24 for most of the list), then large addends
        px = rng.integers(0, 12, (n, 4, 4, 4), dtype=np.uint8)
        px[n - n // 5:] = rng.integers(200, 256, (n // 5, 4, 4, 4), dtype=np.uint8)
    else:                     # one colour: odd values make every add beyond 2^24 a tie of the rounding
        px = np.full((n, 4, 4, 4), 255, np.uint8); px[..., 1] = 129; px[..., 2] = 1
    px[..., 3] = 255
    blocks = np.ascontiguousarray(px)
    perm = rng.permutation(n).astype(np.uint32)
    cuts = [0, n // 2 + 17, n // 2 + 17 + n // 3, n]    # ~524K, ~350K and ~175K texels
    lists = [np.stack([perm[a:b] * 2, perm[a:b] * 2 + 1], axis=1).reshape(-1).astype(np.uint32) for a, b in zip(cuts[:-1], cuts[1:])]
    k = len(lists)
    offs, idx = csr_from_lists(lists)
    d_blocks = hip_ctx.upload(blocks)
    d_offs, d_idx = hip_ctx.upload(offs), hip_ctx.upload(idx)
    params = np.zeros((k, 4), np.uint8); err = np.zeros(k, np.uint64); valid = np.zeros(k, np.uint8)
    oracle().orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), 1, 1, 0, ptr(params), ptr(err, u64p), ptr(valid))
    params1 = params.copy(); params1[0, 0] = min(int(params1[0, 0]) + 1, 31); params1[1, 3] = (int(params1[1, 3]) + 1) % 8
    want1 = (params1.copy(), err.copy(), valid.copy())
    oracle().orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), 1, 1, 1, ptr(want1[0]), ptr(want1[1], u64p), ptr(want1[2]))
    L = hip_ctx.lib
    request.addfinalizer(hip_ctx.set_tuning)
    for wide_min in (32768, 0):   # the many-workgroup passes, then one workgroup per cluster
        hip_ctx.set_tuning(codebook_wide_min=wide_min)
        d_params, d_err, d_valid = hip_ctx.upload(np.zeros((k, 4), np.uint8)), hip_ctx.upload(np.zeros(k, np.uint64)), hip_ctx.upload(np.zeros(k, np.uint8))
        hip_ctx.check(L.k_generate_endpoint_codebook(hip_ctx.h, d_blocks, k, offs.ctypes.data_as(VP), d_offs, d_idx, 1, 1, 0, d_params, d_err, d_valid), "k_gec")
        got = hip_ctx.download(d_params, (k, 4), np.uint8), hip_ctx.download(d_err, (k,), np.uint64), hip_ctx.download(d_valid, (k,), np.uint8)
        assert (got[0] == params).all() and (got[1] == err).all() and (got[2] == valid).all(), (kind, wide_min, got[0].tolist(), params.tolist())
        hip_ctx.check(L.memcpy_h2d(hip_ctx.h, d_params, params1.ctypes.data_as(VP), params1.nbytes), "h2d")
        hip_ctx.check(L.k_generate_endpoint_codebook(hip_ctx.h, d_blocks, k, offs.ctypes.data_as(VP), d_offs, d_idx, 1, 1, 1, d_params, d_err, d_valid), "k_gec")
        got = hip_ctx.download(d_params, (k, 4), np.uint8), hip_ctx.download(d_err, (k,), np.uint64), hip_ctx.download(d_valid, (k,), np.uint8)
        assert (got[0] == want1[0]).all() and (got[1] == want1[1]).all() and (got[2] == want1[2]).all(), (kind, wide_min, "step 1")
        for q in (d_params, d_err, d_valid):
            hip_ctx.free(q)
    for q in (d_blocks, d_offs, d_idx):
        hip_ctx.free(q)


@pytest.mark.parametrize("kind", ["bright", "dark_then_bright", "solid", "ramp", "small"])
def test_ordered_colour_mean_of_clusters(hip_ctx, kind):
    """cbw_ordered_sum (etc1s_codebook_wide.inc) against the thing it replaces, the sequential float loop `fs += (float)value` over the cluster's texels in list order
    (numpy float32 cumulative sums add in order): bit patterns of the three means, for sums that stay exact, that cross one binade, that cross many, whose every add is a tie."""
    rng = np.random.default_rng(11)
    n = {"small": 4096, "ramp": 200000}.get(kind, 65536)
    if kind == "bright":
        px = rng.integers(100, 256, (n, 4, 4, 4), dtype=np.uint8)
    elif kind == "dark_then_bright":
        px = rng.integers(0, 9, (n, 4, 4, 4), dtype=np.uint8); px[n - n // 4:] = rng.integers(180, 256, (n // 4, 4, 4, 4), dtype=np.uint8)
    elif kind == "solid":
        px = np.full((n, 4, 4, 4), 255, np.uint8); px[..., 1] = 129; px[..., 2] = 3
    elif kind == "ramp":      # 3.2 M texels: sums near 4 x 10^8
        px = rng.integers(0, 256, (n, 4, 4, 4), dtype=np.uint8); px[..., 0] |= 0x81
    else:
        px = rng.integers(0, 256, (n, 4, 4, 4), dtype=np.uint8)
    blocks = np.ascontiguousarray(px)
    perm = rng.permutation(n).astype(np.uint32)
    cuts = [0, 3, 3 + n // 7, 3 + n // 7 + n // 3, n]
    lists = [np.stack([perm[a:b] * 2, perm[a:b] * 2 + 1], axis=1).reshape(-1).astype(np.uint32) for a, b in zip(cuts[:-1], cuts[1:])]
    if kind == "small":
        lists[1] = lists[1][:1]   # a cluster of one sub-block
    offs, idx = csr_from_lists(lists)
    d_blocks, d_idx = hip_ctx.upload(blocks), hip_ctx.upload(idx)
    got = np.zeros((len(lists), 3), np.float32)
    hip_ctx.check(hip_ctx.lib.k_cluster_colour_means(hip_ctx.h, d_blocks, len(lists), offs.ctypes.data_as(VP), d_idx, got.ctypes.data_as(VP)), "k_cluster_colour_means")
    for ci, tv in enumerate(lists):
        # texels of training vector block * 2 + subblock: rows 2 * subblock, 2 * subblock + 1 of the block, in raster order (etc.cpp:352-361)
        tex = blocks[tv >> 1].reshape(-1, 2, 8, 4)[np.arange(tv.size), tv & 1].reshape(-1, 4)
        for c in range(3):
            seq = np.cumsum(tex[:, c].astype(np.float32), dtype=np.float32)[-1]   # sequential float32 adds
            want = np.float32(seq) / np.float32(tex.shape[0])
            assert got[ci, c].view(np.uint32) == np.float32(want).view(np.uint32), (kind, ci, c, tex.shape[0], float(got[ci, c]), float(want))
    hip_ctx.free(d_blocks); hip_ctx.free(d_idx)


def _codebook(blocks, k, seed, level=1, perceptual=1):
    rng = np.random.default_rng(seed)
    lists, block_cluster = _clusters_by_luma(blocks, k, rng)
    offs, idx = csr_from_lists(lists)
    params = np.zeros((k, 4), np.uint8); err = np.zeros(k, np.uint64); valid = np.zeros(k, np.uint8)
    oracle().orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), level, perceptual, 0, ptr(params), ptr(err, u64p), ptr(valid))
    return params, block_cluster


@pytest.mark.parametrize("presorted", [True, False])
@pytest.mark.parametrize("perceptual", [1, 0])
@pytest.mark.parametrize("hier", [True, False, "foreign"])
def test_refine_endpoint_clusterization(hip_ctx, blocks, d_blocks, hier, perceptual, presorted, request):
    # presorted: the candidate lists re-sorted on the device by (clamping class, intensity table) first (k_refine_sort_lists + k_refine_sorted, the
    # default); otherwise the kernel that filters and classifies the lists as it reads them (kept for codebooks beyond 65,535 entries)
    if not presorted:
        hip_ctx.set_tuning(refine_unsorted=1)
        request.addfinalizer(hip_ctx.set_tuning)   # back to the process defaults
    n = blocks.shape[0]
    k = 300
    params, block_cluster = _codebook(blocks, k, 11, perceptual=perceptual)
    # duplicate a few codebook entries so that ties between distinct clusters occur
    params[10] = params[11]; params[200] = params[150]
    n_parents = 7 if hier else 0
    if hier:
        # parents = contiguous ranges of clusters (luma ordered), candidate lists = clusters of that parent, sorted ascending like
        # compute_endpoint_clusters_within_each_parent_cluster (frontend.cpp:971-996)
        cluster_parent = (np.arange(k) * n_parents // k).astype(np.uint8)
        block_parent = cluster_parent[block_cluster]
        if hier == "foreign":
            # blocks filed under a parent whose list does NOT hold their current cluster: never produced by the frontend, but the kernel's pruning
            # threshold (the current cluster's error) is only valid for list members, so this exercises its fall-back
            block_parent = np.random.default_rng(5).integers(0, n_parents, n).astype(np.uint8)
        cand = [np.nonzero(cluster_parent == p)[0].astype(np.uint32) for p in range(n_parents)]
        coffs, cidx = csr_from_lists(cand)
    else:
        block_parent = np.zeros(n, np.uint8); coffs = np.zeros(1, np.uint32); cidx = np.zeros(1, np.uint32)
    exp = np.zeros(n, np.uint32)
    oracle().orc_refine_endpoint_clusterization(ptr(blocks), n, ptr(block_cluster, u32p), ptr(params), k, n_parents, ptr(coffs, u32p), ptr(cidx, u32p),
                                                ptr(block_parent), perceptual, ptr(exp, u32p))
    bufs = [hip_ctx.upload(a) for a in (block_cluster, params, coffs, cidx, block_parent)]
    d_out = hip_ctx.alloc(n * 4)
    hip_ctx.check(hip_ctx.lib.k_refine_endpoint_clusterization(hip_ctx.h, d_blocks, n, bufs[0], bufs[1], k, n_parents, bufs[2], bufs[3], bufs[4], perceptual, d_out), "k_refine")
    got = hip_ctx.download(d_out, (n,), np.uint32)
    assert (got == exp).all(), f"{int((got != exp).sum())} of {n} differ"
    assert (got != block_cluster).any(), "degenerate test: nothing moved"
    for p in bufs + [d_out]:
        hip_ctx.free(p)


@pytest.mark.parametrize("perceptual", [1, 0])
def test_determine_selectors(hip_ctx, blocks, d_blocks, perceptual):
    n = blocks.shape[0]
    params, block_cluster = _codebook(blocks, 64, 3, perceptual=perceptual)
    per_block = np.ascontiguousarray(params[block_cluster])
    exp = np.zeros((n, 8), np.uint8)
    oracle().orc_determine_selectors(ptr(blocks), n, ptr(per_block), perceptual, ptr(exp))
    d_out = hip_ctx.alloc(n * 8)
    # (a) per-block colour table, as the reference seam passes it
    d_pb = hip_ctx.upload(per_block)
    hip_ctx.check(hip_ctx.lib.k_determine_selectors(hip_ctx.h, d_blocks, n, d_pb, None, perceptual, d_out), "k_ds")
    assert (hip_ctx.download(d_out, (n, 8), np.uint8) == exp).all()
    # (b) codebook + per-block cluster index, as the resident frontend calls it
    d_par, d_bc = hip_ctx.upload(params), hip_ctx.upload(block_cluster)
    hip_ctx.check(hip_ctx.lib.memset(hip_ctx.h, d_out, 0, n * 8), "memset")
    hip_ctx.check(hip_ctx.lib.k_determine_selectors(hip_ctx.h, d_blocks, n, d_par, d_bc, perceptual, d_out), "k_ds")
    assert (hip_ctx.download(d_out, (n, 8), np.uint8) == exp).all()
    for p in (d_out, d_pb, d_par, d_bc):
        hip_ctx.free(p)


def _encoded(blocks, perceptual=1):
    n = blocks.shape[0]
    params, block_cluster = _codebook(blocks, 64, 3, perceptual=perceptual)
    enc = np.zeros((n, 8), np.uint8)
    oracle().orc_determine_selectors(ptr(blocks), n, ptr(np.ascontiguousarray(params[block_cluster])), perceptual, ptr(enc))
    return enc


@pytest.mark.parametrize("perceptual", [1, 0])
def test_selector_training_vectors(hip_ctx, blocks, perceptual):
    n = blocks.shape[0]
    enc = _encoded(blocks, perceptual)
    exp_v = np.zeros((n, 16), np.float32); exp_w = np.zeros(n, np.uint64)
    oracle().orc_selector_training_vectors(ptr(enc), n, perceptual, ptr(exp_v, f32p), ptr(exp_w, u64p))
    d_enc = hip_ctx.upload(enc); d_v = hip_ctx.alloc(n * 64); d_w = hip_ctx.alloc(n * 8)
    hip_ctx.check(hip_ctx.lib.k_selector_training_vectors(hip_ctx.h, d_enc, n, perceptual, d_v, d_w), "k_stv")
    assert (hip_ctx.download(d_v, (n, 16), np.float32) == exp_v).all()
    assert (hip_ctx.download(d_w, (n,), np.uint64) == exp_w).all()
    for p in (d_enc, d_v, d_w):
        hip_ctx.free(p)


def _check_groups(n, u, keys_np, got_keys, got_offsets, got_idx, weights=None, got_weights=None):
    """distinct keys ascending; blocks of every distinct vector ascending and complete; summed weights"""
    uniq, inverse = np.unique(keys_np, return_inverse=True)
    assert u == uniq.size and (got_keys[:u] == uniq).all()
    assert got_offsets[0] == 0 and got_offsets[u] == n and (np.diff(got_offsets[:u + 1].astype(np.int64)) > 0).all()
    order = np.argsort(keys_np, kind="stable")        # stable: ascending block index inside a key
    assert (got_idx == order).all()
    if weights is not None:
        assert (got_weights[:u] == np.array([weights[order[got_offsets[i]:got_offsets[i + 1]]].sum() for i in range(u)], np.uint64)).all()


@pytest.mark.parametrize("n", [0, 1, 2, 257, 3072])
def test_unique_selector_vectors(hip_ctx, blocks, n):
    """bu_hip_k_unique_selector_vectors against the oracle's selector training vectors + numpy: the key is the vector's 16 values, first in the
    top two bits (the reference's std::map<vec16F> order); all-equal and all-distinct inputs included through the block sample."""
    enc_all = _encoded(blocks, 1)
    enc = np.ascontiguousarray(np.concatenate([enc_all, enc_all[:64]])[:n])   # duplicates guaranteed
    vec = np.zeros((n, 16), np.float32); w = np.zeros(n, np.uint64)
    if n:
        oracle().orc_selector_training_vectors(ptr(enc), n, 1, ptr(vec, f32p), ptr(w, u64p))
    keys = (vec.astype(np.uint32) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32) if n else np.zeros(0, np.uint32)
    d_enc, d_w = hip_ctx.upload(enc), hip_ctx.upload(w)
    d_idx, d_keys, d_uw, d_ofs = hip_ctx.alloc(n * 4 + 4), hip_ctx.alloc(n * 4 + 4), hip_ctx.alloc(n * 8 + 8), hip_ctx.alloc(n * 4 + 8)
    u = C.c_uint32(12345)
    hip_ctx.check(hip_ctx.lib.k_unique_selector_vectors(hip_ctx.h, d_enc, d_w, n, d_idx, d_keys, d_uw, d_ofs, C.byref(u)), "k_unique_selector_vectors")
    if n:
        _check_groups(n, u.value, keys, hip_ctx.download(d_keys, (n,), np.uint32), hip_ctx.download(d_ofs, (n + 1,), np.uint32),
                      hip_ctx.download(d_idx, (n,), np.uint32), w, hip_ctx.download(d_uw, (n,), np.uint64))
    else:
        assert u.value == 0
    for p_ in (d_enc, d_w, d_idx, d_keys, d_uw, d_ofs):
        hip_ctx.free(p_)


@pytest.mark.parametrize("n", [0, 1, 2, 257, 3072])
def test_unique_endpoint_vectors(hip_ctx, blocks, n):
    """bu_hip_k_unique_endpoint_vectors against the oracle's endpoint training vectors: key bytes / 255 are the vec6F, order is lexicographic."""
    etc_all = _encode_ref(blocks)
    etc = np.ascontiguousarray(np.concatenate([etc_all, etc_all[:64]])[:n])
    vec = np.zeros((n, 6), np.float32)
    if n:
        oracle().orc_endpoint_training_vectors(ptr(etc), n, ptr(vec, f32p))
    bytes_ = np.rint(vec.astype(np.float64) * 255.0).astype(np.uint64)
    assert (bytes_.astype(np.float32) * np.float32(1.0 / 255.0) == vec).all()   # the float the reference stores is byte * (1/255)
    keys = (bytes_ << (8 * (5 - np.arange(6, dtype=np.uint64)))).sum(axis=1).astype(np.uint64) if n else np.zeros(0, np.uint64)
    d_etc = hip_ctx.upload(etc)
    d_idx, d_keys, d_ofs = hip_ctx.alloc(n * 4 + 4), hip_ctx.alloc(n * 8 + 8), hip_ctx.alloc(n * 4 + 8)
    u = C.c_uint32(12345)
    hip_ctx.check(hip_ctx.lib.k_unique_endpoint_vectors(hip_ctx.h, d_etc, n, d_idx, d_keys, d_ofs, C.byref(u)), "k_unique_endpoint_vectors")
    if n:
        _check_groups(n, u.value, keys, hip_ctx.download(d_keys, (n,), np.uint64), hip_ctx.download(d_ofs, (n + 1,), np.uint32), hip_ctx.download(d_idx, (n,), np.uint32))
    else:
        assert u.value == 0
    for p_ in (d_etc, d_idx, d_keys, d_ofs):
        hip_ctx.free(p_)


def _selector_clusters(enc, k, rng):
    n = enc.shape[0]
    key = enc[:, 4:].astype(np.uint32) @ np.array([1, 256, 65536, 16777216], np.uint32)
    order = np.argsort(key, kind="stable")
    cuts = np.sort(rng.choice(np.arange(1, n), size=k - 2, replace=False))
    lists = [rng.permutation(r).astype(np.uint32) for r in np.split(order, cuts)]
    lists.insert(5, np.zeros(0, np.uint32))  # an empty cluster (frontend.cpp:2282-2283)
    return lists


@pytest.mark.parametrize("perceptual", [1, 0])
def test_create_optimized_selector_codebook(hip_ctx, blocks, d_blocks, perceptual):
    enc = _encoded(blocks, perceptual)
    rng = np.random.default_rng(5)
    k = 120
    lists = _selector_clusters(enc, k, rng)
    offs, idx = csr_from_lists(lists)
    sel = rng.integers(0, 256, (k, 8), dtype=np.uint8)  # stale contents: only the 4 selector bytes of non-empty clusters may change
    exp = sel.copy()
    oracle().orc_create_optimized_selector_codebook(ptr(blocks), ptr(enc), k, ptr(offs, u32p), ptr(idx, u32p), perceptual, ptr(exp))
    bufs = [hip_ctx.upload(a) for a in (enc, offs, idx, sel)]
    hip_ctx.check(hip_ctx.lib.k_create_optimized_selector_codebook(hip_ctx.h, d_blocks, bufs[0], k, bufs[1], bufs[2], perceptual, bufs[3]), "k_cosc")
    got = hip_ctx.download(bufs[3], (k, 8), np.uint8)
    assert (got == exp).all()
    for p in bufs:
        hip_ctx.free(p)


@pytest.mark.parametrize("perceptual", [1, 0])
@pytest.mark.parametrize("hier", [True, False])
def test_find_optimal_selector_clusters(hip_ctx, blocks, hier, perceptual):
    # make runs of identical tiles that straddle a job boundary (chunk) so the shortcut of frontend.cpp:2557-2564 is exercised
    blocks = blocks.copy()
    blocks[100:110] = blocks[100]
    blocks[2040:2060] = blocks[2040]
    n = blocks.shape[0]
    enc = _encoded(blocks, perceptual)
    enc[105, :3] ^= 0x18  # identical tiles with DIFFERENT endpoints: the copied choice must still be the run head's
    rng = np.random.default_rng(9)
    k = 200
    lists = _selector_clusters(enc, k, rng)
    offs, idx = csr_from_lists(lists)
    sel = np.zeros((k, 8), np.uint8)
    oracle().orc_create_optimized_selector_codebook(ptr(blocks), ptr(enc), k, ptr(offs, u32p), ptr(idx, u32p), perceptual, ptr(sel))
    n_parents = 5 if hier else 0
    if hier:
        cluster_parent = (np.arange(k) * n_parents // k).astype(np.uint8)
        block_sel = np.zeros(n, np.uint32)
        for ci, l in enumerate(lists):
            block_sel[l] = ci
        block_parent = cluster_parent[block_sel]
        coffs, cidx = csr_from_lists([np.nonzero(cluster_parent == p)[0].astype(np.uint32) for p in range(n_parents)])
    else:
        block_parent = np.zeros(n, np.uint8); coffs = np.zeros(1, np.uint32); cidx = np.zeros(1, np.uint32)
    exp_enc = enc.copy(); exp_idx = np.zeros(n, np.uint32)
    oracle().orc_find_optimal_selector_clusters(ptr(blocks), ptr(exp_enc), n, ptr(sel), k, n_parents, ptr(coffs, u32p), ptr(cidx, u32p), ptr(block_parent),
                                                perceptual, 2048, ptr(exp_idx, u32p))
    bufs = [hip_ctx.upload(a) for a in (blocks, enc, sel, coffs, cidx, block_parent)]
    d_out = hip_ctx.alloc(n * 4)
    hip_ctx.check(hip_ctx.lib.k_find_optimal_selector_clusters(hip_ctx.h, bufs[0], bufs[1], n, bufs[2], k, n_parents, bufs[3], bufs[4], bufs[5], perceptual, 2048, d_out), "k_fosc")
    got_idx = hip_ctx.download(d_out, (n,), np.uint32)
    got_enc = hip_ctx.download(bufs[1], (n, 8), np.uint8)
    assert (got_idx == exp_idx).all(), f"{int((got_idx != exp_idx).sum())} differ"
    assert (got_enc == exp_enc).all()
    for p in bufs + [d_out]:
        hip_ctx.free(p)


@pytest.mark.parametrize("w,h,pad", [(64, 64, 0), (130, 67, 0), (5, 3, 0), (1, 1, 0), (257, 129, 12), (4096, 2048, 0)])
def test_extract_blocks(hip_ctx, w, h, pad):
    """extract_source_blocks on the device (comp.cpp:3207-3268): tiles of an RGBA raster, edges clamped, arbitrary row pitch."""
    rng = np.random.default_rng(w * 1000 + h)
    pitch = w * 4 + pad
    raster = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    img = np.ascontiguousarray(raster[:, :w * 4].reshape(h, w, 4))
    want = to_pixel_blocks(img)
    d_img = hip_ctx.upload(raster)
    d_out = hip_ctx.alloc(want.size)
    hip_ctx.check(hip_ctx.lib.k_extract_blocks(hip_ctx.h, d_img, w, h, pitch, d_out), "k_extract_blocks")
    got = hip_ctx.download(d_out, want.shape, np.uint8)
    hip_ctx.free(d_img); hip_ctx.free(d_out)
    assert (got == want).all()
    assert hip_ctx.lib.k_extract_blocks(hip_ctx.h, d_img, w, h, w * 4 - 1, d_out) == 0  # pitch smaller than a row: refused, not a crash


def test_two_contexts_on_two_threads(hip_ctx):
    """The reference's threading rule (basisu_opencl.h:28-31): one context per thread, different contexts may run concurrently."""
    import threading
    from basis_universal_amd import capi, uastc
    from basis_universal_amd.etc1s import Etc1sFrontend
    imgs = [to_pixel_blocks(synth(192, 128, 100 + i)) for i in range(2)]

    def encode(ctx, blocks):
        fe = Etc1sFrontend(ctx)
        fe.init(blocks, 200, 200, 1, True)
        fe.compress()
        out = (fe.get("encoded_blocks").copy(), uastc.encode_uastc_blocks(ctx, blocks, 2))
        fe.close()
        return out

    serial = [encode(hip_ctx, b) for b in imgs]
    results, errors = [None, None], []

    def worker(i):
        try:
            ctx = capi.Context(0)
            for _ in range(3):
                results[i] = encode(ctx, imgs[i])
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for i in range(2):
        assert (results[i][0] == serial[i][0]).all() and (results[i][1] == serial[i][1]).all()


def test_error_convention(hip_ctx):
    """Failures return 0 and leave a message; nothing throws or aborts (opencl.cpp:972-976 convention)."""
    lib = hip_ctx.lib
    assert lib.k_encode_uastc_blocks(hip_ctx.h, None, 16, 2, None) == 0 and b"null" in lib.dll.bu_hip_last_error(hip_ctx.h)
    assert lib.encode_uastc_blocks(None, None, 2) == 0
    assert lib.k_generate_endpoint_codebook_part(hip_ctx.h, None, 4, None, None, None, 1, 1, 0, None, None, None, 3, 2) == 0
