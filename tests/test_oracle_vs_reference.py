"""Pins the plain-C oracle (oracle/etc1s_oracle.c) to the REAL reference (oracle/_ref/libref_harness.so, compiled from
/root/reference by oracle/Makefile): every oracle function is fed the reference frontend's own intermediate state and must
reproduce the reference's next state bit for bit. CPU only; skipped where the reference build is not present.
"""
import numpy as np
import pytest

from helpers import (oracle, ref, have_ref, RefFrontend, ptr, u32p, u64p, f32p, synth, uniform_random, to_pixel_blocks,
                     load_png, REF_DIR, csr_from_lists)

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def _inputs():
    out = {"synth": to_pixel_blocks(synth(256, 192, 1234)), "noise": to_pixel_blocks(uniform_random(96, 64, 42))}
    k = REF_DIR / "test_files" / "kodim03.png"
    if k.exists():
        out["kodim03"] = to_pixel_blocks(load_png(k))
    return out


INPUTS = _inputs()


@pytest.mark.parametrize("name", sorted(INPUTS))
@pytest.mark.parametrize("level,perceptual", [(0, 1), (1, 1), (1, 0), (2, 1), (6, 0)])
def test_block_encode(name, level, perceptual):
    blocks = INPUTS[name]
    if level >= 2:
        blocks = blocks[:2048]
    n = blocks.shape[0]
    a = np.zeros((n, 8), np.uint8); b = np.zeros((n, 8), np.uint8)
    oracle().orc_encode_etc1s_blocks(ptr(blocks), n, level, perceptual, ptr(a))
    ref().ref_encode_etc1s_blocks(ptr(blocks), n, level, perceptual, ptr(b))
    assert (a == b).all()


def test_color_distance_and_hash_exhaustive_sample():
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (20000, 2, 3), dtype=np.uint8)
    px[:64, 0] = [[0, 0, 0]]; px[:64, 1] = [[255, 255, 255]]
    px[64:128, 0] = [[255, 0, 255]]; px[64:128, 1] = [[0, 255, 0]]
    O, R = oracle(), ref()
    for perceptual in (0, 1):
        for a, b in px[:4000]:
            a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
            assert O.orc_color_distance(perceptual, ptr(a), ptr(b)) == R.ref_color_distance(perceptual, ptr(a), ptr(b))


@pytest.mark.parametrize("quality", [1, 2, 3])
@pytest.mark.parametrize("n", [8, 16, 24, 1000, 70000])
def test_cluster_optimizer(n, quality):
    """etc1_optimizer on pixel lists longer than a block, including one whose channel sums exceed 2^24 (float mean, H4)."""
    rng = np.random.default_rng(n + quality)
    base = rng.integers(0, 256, 3)
    px = np.clip(base[None, :] + rng.normal(0, 25 if n < 70000 else 3, (n, 3)), 0, 255).astype(np.uint8)
    if n == 70000:
        px[:, 1] = 250  # sum = 17.5M > 2^24
    rgba = np.concatenate([px, np.full((n, 1), 255, np.uint8)], axis=1)
    rgba = np.ascontiguousarray(rgba)
    for perceptual in (1, 0):
        ca = np.zeros(3, np.uint8); cb = np.zeros(3, np.uint8)
        ia = np.zeros(1, np.uint32); ib = np.zeros(1, np.uint32)
        ea = np.zeros(1, np.uint64); eb = np.zeros(1, np.uint64)
        sa = np.zeros(n, np.uint8); sb = np.zeros(n, np.uint8)
        assert oracle().orc_etc1_optimize(ptr(rgba), n, quality, perceptual, ptr(ca), ptr(ia, u32p), ptr(ea, u64p), ptr(sa)) == 1
        assert ref().ref_etc1_optimize(ptr(rgba), n, quality, perceptual, ptr(cb), ptr(ib, u32p), ptr(eb, u64p), ptr(sb)) == 1
        assert (ca == cb).all() and ia[0] == ib[0] and ea[0] == eb[0] and (sa == sb).all()


def _lists(offs, idx):
    return [idx[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


@pytest.mark.parametrize("name,level,perceptual,max_ep,max_sel", [
    ("synth", 1, 1, 400, 500), ("synth", 2, 0, 300, 300), ("noise", 1, 1, 64, 64), ("kodim03", 1, 1, 2416, 2731), ("synth", 3, 1, 300, 400),
])
def test_frontend_stages(name, level, perceptual, max_ep, max_sel):
    if name not in INPUTS:
        pytest.skip("kodim03 not available")
    blocks = INPUTS[name]
    n = blocks.shape[0]
    O = oracle()
    fe = RefFrontend(blocks, max_ep, max_sel, level, perceptual)
    hier_ep = bool(fe.call("use_hierarchical_endpoint_codebooks"))
    hier_sel = bool(fe.call("use_hierarchical_selector_codebooks"))

    # a6
    fe.call("init_etc1_images")
    etc1 = fe.get("etc1_blocks").reshape(n, 8)
    mine = np.zeros((n, 8), np.uint8)
    O.orc_encode_etc1s_blocks(ptr(blocks), n, level, perceptual, ptr(mine))
    assert (mine == etc1).all()

    # a7
    fe.call("init_endpoint_training_vectors")
    tv = fe.get("endpoint_training_vecs").reshape(2 * n, 32)
    tv_f = tv[:, :24].copy().view(np.float32).reshape(2 * n, 6)
    v6 = np.zeros((n, 6), np.float32)
    O.orc_endpoint_training_vectors(ptr(etc1), n, ptr(v6, f32p))
    assert (tv_f[0::2].view(np.uint32) == v6.view(np.uint32)).all() and (tv_f[1::2].view(np.uint32) == v6.view(np.uint32)).all()
    assert (tv[:, 24:].copy().view(np.uint64) == 1).all()

    # a8 (reference TSVQ) then a9
    fe.call("generate_endpoint_clusters")
    offs, idx = fe.get_csr("endpoint_clusters")
    k = len(offs) - 1
    fe.call("generate_endpoint_codebook", 0)
    prm = fe.get("endpoint_cluster_etc_params").reshape(k, 16)
    params = np.zeros((k, 4), np.uint8); err = np.zeros(k, np.uint64); valid = np.zeros(k, np.uint8)
    O.orc_generate_endpoint_codebook(ptr(blocks), k, ptr(offs, u32p), ptr(idx, u32p), level, perceptual, 0, ptr(params), ptr(err, u64p), ptr(valid))
    assert (params == prm[:, :4]).all()
    assert (err == prm[:, 8:].copy().view(np.uint64).reshape(k)).all()

    # a10
    block_cluster = np.zeros(n, np.uint32)
    for ci, l in enumerate(_lists(offs, idx)):
        block_cluster[l >> 1] = ci
    fe.call("refine_endpoint_clusterization")
    offs2, idx2 = fe.get_csr("endpoint_clusters")
    new_cluster = np.zeros(n, np.uint32)
    for ci, l in enumerate(_lists(offs2, idx2)):
        new_cluster[l >> 1] = ci
    if hier_ep:
        coffs, cidx = fe.get_csr("endpoint_clusters_within_each_parent_cluster")
        bparent = fe.get("block_parent_endpoint_cluster")
        npar = len(coffs) - 1
    else:
        coffs = np.zeros(1, np.uint32); cidx = np.zeros(1, np.uint32); bparent = np.zeros(n, np.uint8); npar = 0
    best = np.zeros(n, np.uint32)
    O.orc_refine_endpoint_clusterization(ptr(blocks), n, ptr(block_cluster, u32p), ptr(params), k, npar, ptr(coffs, u32p), ptr(cidx, u32p),
                                         ptr(bparent), perceptual, ptr(best, u32p))
    assert (best == new_cluster).all()

    # host bookkeeping of the reference, then a11
    fe.call("eliminate_redundant_or_empty_endpoint_clusters")
    fe.call("generate_block_endpoint_clusters")
    fe.call("create_initial_packed_texture")
    enc = fe.get("encoded_blocks").reshape(n, 8)
    bci = fe.get("block_endpoint_clusters_indices", np.uint32)
    prm2 = fe.get("endpoint_cluster_etc_params").reshape(-1, 16)
    per_block = np.ascontiguousarray(prm2[bci][:, :4])
    mine = np.zeros((n, 8), np.uint8)
    O.orc_determine_selectors(ptr(blocks), n, ptr(per_block), perceptual, ptr(mine))
    assert (mine == enc).all()

    # a8 again (reference TSVQ on selectors), then a13, a14
    fe.call("generate_selector_clusters")
    if hier_sel:
        fe.call("compute_selector_clusters_within_each_parent_cluster")
    soffs, sidx = fe.get_csr("selector_cluster_block_indices")
    ks = len(soffs) - 1
    fe.call("create_optimized_selector_codebook", 0)
    sel = fe.get("optimized_cluster_selectors").reshape(ks, 8)
    mine_sel = np.zeros((ks, 8), np.uint8)
    O.orc_create_optimized_selector_codebook(ptr(blocks), ptr(enc), ks, ptr(soffs, u32p), ptr(sidx, u32p), perceptual, ptr(mine_sel))
    # only the four selector bytes of an entry are ever meaningful (the reference leaves the colour bytes uninitialised)
    assert (mine_sel[:, 4:] == sel[:, 4:]).all()

    if level >= 1:
        if hier_sel:
            scoffs, scidx = fe.get_csr("selector_clusters_within_each_parent_cluster")
            sparent = fe.get("block_parent_selector_cluster")
            nspar = len(scoffs) - 1
        else:
            scoffs = np.zeros(1, np.uint32); scidx = np.zeros(1, np.uint32); sparent = np.zeros(n, np.uint8); nspar = 0
        fe.call("find_optimal_selector_clusters_for_each_block")
        enc2 = fe.get("encoded_blocks").reshape(n, 8)
        bsi = fe.get("block_selector_cluster_index", np.uint32)
        mine_enc = enc.copy(); mine_idx = np.zeros(n, np.uint32)
        O.orc_find_optimal_selector_clusters(ptr(blocks), ptr(mine_enc), n, ptr(sel), ks, nspar, ptr(scoffs, u32p), ptr(scidx, u32p), ptr(sparent),
                                             perceptual, 2048, ptr(mine_idx, u32p))
        assert (mine_idx == bsi).all()
        assert (mine_enc == enc2).all()
    fe.close()
