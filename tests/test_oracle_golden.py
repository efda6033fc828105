"""CPU: the C oracle against the committed known-answer vectors (tests/golden/etc1s_reference_vectors.npz), which were produced
by the real reference build (tools/gen_golden_oracle.py). Runs everywhere, including where oracle/_ref is absent."""
import pathlib

import numpy as np

from helpers import oracle, ptr, u32p, u64p

G = np.load(pathlib.Path(__file__).parent / "golden" / "etc1s_reference_vectors.npz")


def test_block_encode_known_answers():
    blocks = np.ascontiguousarray(G["blocks"])
    n = blocks.shape[0]
    for level in (0, 1, 2, 6):
        for perc in (0, 1):
            out = np.zeros((n, 8), np.uint8)
            oracle().orc_encode_etc1s_blocks(ptr(blocks), n, level, perc, ptr(out))
            assert (out == G[f"etc1s_l{level}_p{perc}"]).all(), (level, perc)


def test_cluster_optimizer_known_answers():
    for row in G["cluster_results"]:
        i, q, perc, r, g, b, inten, err = [int(v) for v in row]
        rgba = np.ascontiguousarray(G[f"cluster_px_{i}"])
        c = np.zeros(3, np.uint8); it = np.zeros(1, np.uint32); e = np.zeros(1, np.uint64)
        assert oracle().orc_etc1_optimize(ptr(rgba), rgba.shape[0], q, perc, ptr(c), ptr(it, u32p), ptr(e, u64p), None) == 1
        assert (int(c[0]), int(c[1]), int(c[2]), int(it[0]), int(e[0])) == (r, g, b, inten, err)


def test_frontend_fixture_is_self_consistent():
    """The reference frontend's final blocks are exactly determine_selectors/selector stamping of its codebooks."""
    blocks = np.ascontiguousarray(G["blocks"]); n = blocks.shape[0]
    prm = G["fe_endpoint_cluster_etc_params"].reshape(-1, 16)
    bci = G["fe_block_endpoint_clusters_indices"].view(np.uint32)
    enc = G["fe_encoded_blocks"].reshape(n, 8)
    sel = G["fe_optimized_cluster_selectors"].reshape(-1, 8)
    bsi = G["fe_block_selector_cluster_index"].view(np.uint32)
    per_block = np.ascontiguousarray(prm[bci][:, :4])
    mine = np.zeros((n, 8), np.uint8)
    oracle().orc_determine_selectors(ptr(blocks), n, ptr(per_block), 1, ptr(mine))
    assert (mine[:, :4] == enc[:, :4]).all()           # colour/inten/flags bytes
    assert (enc[:, 4:] == sel[bsi][:, 4:]).all()       # selector bytes come from the selector codebook
