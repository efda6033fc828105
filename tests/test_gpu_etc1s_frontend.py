"""-m gpu end-to-end parity of the resident ETC1S frontend (bu::etc1s_frontend over the HIP kernels) against the REAL reference
frontend (oracle/_ref, single-threaded = the pinned configuration, SURVEY hazard H1) where it is available, and against the
committed golden digests (tests/golden/etc1s_frontend_digests.json, produced by that same reference build) everywhere.
Everything compared is integer data; the comparison is exact.
"""
import hashlib
import json
import pathlib

import numpy as np
import pytest

from helpers import have_ref, RefFrontend, synth, uniform_random, to_pixel_blocks, load_png, REF_DIR

pytestmark = pytest.mark.gpu

GOLDEN = pathlib.Path(__file__).parent / "golden" / "etc1s_frontend_digests.json"

CASES = {
    # name: (image factory, max_ep, max_sel, level, perceptual)
    "synth256_l1": (lambda: synth(256, 192, 1234), 400, 500, 1, True),
    "synth256_l2_linear": (lambda: synth(256, 192, 1234), 300, 300, 2, False),
    "synth256_l3_flat": (lambda: synth(256, 192, 77), 300, 400, 3, True),
    "synth256_l0": (lambda: synth(256, 192, 5), 300, 300, 0, True),
    "noise_small_codebooks": (lambda: uniform_random(96, 64, 42), 64, 64, 1, True),
    "synth512_q128": (lambda: synth(512, 512, 99), None, None, 1, True),
    "ragged_edges": (lambda: synth(132, 68, 3)[:67, :130], 128, 128, 1, True),
    # levels 4-6: several endpoint / selector iterations, new clusters introduced between them, endpoints refitted to the selectors (row a15)
    "synth256_l4": (lambda: synth(256, 192, 21), 300, 300, 4, True),
    "synth256_l5_linear": (lambda: synth(192, 128, 22), 200, 256, 5, False),
    "synth128_l6": (lambda: synth(128, 128, 23), 128, 160, 6, True),
    "noise_l4": (lambda: uniform_random(96, 64, 7), 100, 100, 4, True),
}

STATE = ["etc1_blocks", "endpoint_cluster_etc_params", "block_endpoint_clusters_indices", "orig_encoded_blocks", "encoded_blocks",
         "optimized_cluster_selectors", "block_selector_cluster_index", "endpoint_clusters", "selector_cluster_block_indices"]


def _canon(name, arr):
    a = np.asarray(arr)
    if name == "optimized_cluster_selectors":
        return np.ascontiguousarray(a.reshape(-1, 8)[:, 4:])  # colour bytes of codebook entries are never meaningful
    if name == "endpoint_cluster_etc_params":
        a = a.reshape(-1, 16).copy(); a[:, 4:8] = 0           # keep r,g,b,inten and the u64 error; drop the flag padding
        return a
    return np.ascontiguousarray(a)


def _digest(state):
    return {k: hashlib.sha256(_canon(k, v).tobytes()).hexdigest() for k, v in state.items()}


def _run_hip(hip_ctx, blocks, max_ep, max_sel, level, perceptual):
    from basis_universal_amd.etc1s import Etc1sFrontend
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, level, perceptual)
    fe.compress()
    st = {k: fe.get(k) for k in STATE}
    fe.close()
    return st


def _params(case):
    from basis_universal_amd.etc1s import quality_to_clusters
    img_fn, max_ep, max_sel, level, perceptual = CASES[case]
    blocks = to_pixel_blocks(img_fn())
    if max_ep is None:
        max_ep, max_sel = quality_to_clusters(128, blocks.shape[0])
    return blocks, max_ep, max_sel, level, perceptual


@pytest.mark.parametrize("case", sorted(CASES))
def test_frontend_matches_golden(hip_ctx, case):
    golden = json.loads(GOLDEN.read_text())
    blocks, max_ep, max_sel, level, perceptual = _params(case)
    got = _digest(_run_hip(hip_ctx, blocks, max_ep, max_sel, level, perceptual))
    assert got == golden[case]["digests"], {k: (got[k][:12], golden[case]["digests"][k][:12]) for k in got if got[k] != golden[case]["digests"][k]}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("case", ["synth256_l1", "synth256_l3_flat", "noise_small_codebooks", "synth256_l4", "synth256_l5_linear", "synth128_l6", "noise_l4"])
def test_frontend_matches_live_reference(hip_ctx, case):
    blocks, max_ep, max_sel, level, perceptual = _params(case)
    got = _run_hip(hip_ctx, blocks, max_ep, max_sel, level, perceptual)
    fe = RefFrontend(blocks, max_ep, max_sel, level, perceptual)
    fe.call("compress")
    for k in STATE:
        exp = fe.get(k)
        assert (_canon(k, got[k]) == _canon(k, exp)).all() if _canon(k, got[k]).shape == _canon(k, exp).shape else False, k
    fe.close()


@pytest.mark.skipif(not (have_ref() and (REF_DIR / "test_files" / "kodim03.png").exists()), reason="needs /root/reference")
def test_kodim03_q128_matches_live_reference(hip_ctx):
    """BASELINE config #1: kodim03, -q 128, CLI comp level 1."""
    from basis_universal_amd.etc1s import quality_to_clusters
    blocks = to_pixel_blocks(load_png(REF_DIR / "test_files" / "kodim03.png"))
    max_ep, max_sel = quality_to_clusters(128, blocks.shape[0])
    assert (max_ep, max_sel) == (2416, 2731)
    got = _run_hip(hip_ctx, blocks, max_ep, max_sel, 1, True)
    fe = RefFrontend(blocks, max_ep, max_sel, 1, True)
    fe.call("compress")
    for k in STATE:
        assert (_canon(k, got[k]) == _canon(k, fe.get(k))).all(), k
