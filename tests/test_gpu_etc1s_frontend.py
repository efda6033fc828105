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


def _sorted_lists(blob):
    """A CSR blob (count, offsets, indices) with every list sorted: the membership of the clusters without the order inside them."""
    u = np.ascontiguousarray(blob).view(np.uint32).copy()
    n = int(u[0])
    offs, idx = u[1:n + 2], u[n + 2:]
    for i in range(n):
        idx[offs[i]:offs[i + 1]].sort()
    return u


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


@pytest.mark.parametrize("case", ["synth256_l1", "synth256_l2_linear", "synth256_l4", "synth256_l5_linear", "synth128_l6", "noise_l4"])
def test_frontend_with_every_codebook_fit_through_the_many_workgroup_passes(hip_ctx, case, request):
    """bu_hip_tuning::codebook_wide_min = 8: EVERY endpoint cluster is fitted by etc1s_codebook_wide.inc -- generate_endpoint_codebook at step 0 and step > 0, and from
    level 4 up the fit with forced selectors (refine_block_endpoints_given_selectors) -- instead of the large ones only. Same digests as the reference."""
    request.addfinalizer(hip_ctx.set_tuning)
    hip_ctx.set_tuning(codebook_wide_min=8)
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


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_kodim03_q128_matches_live_reference(hip_ctx):
    """BASELINE config #1: kodim03, -q 128, CLI comp level 1 (pixels from tests/golden/kodim03.npz: the GPU box has no /root/reference/test_files)."""
    from basis_universal_amd.etc1s import quality_to_clusters
    blocks = to_pixel_blocks(np.ascontiguousarray(np.load(pathlib.Path(__file__).parent / "golden" / "kodim03.npz")["rgba"]))
    max_ep, max_sel = quality_to_clusters(128, blocks.shape[0])
    assert (max_ep, max_sel) == (2416, 2731)
    got = _run_hip(hip_ctx, blocks, max_ep, max_sel, 1, True)
    fe = RefFrontend(blocks, max_ep, max_sel, 1, True)
    fe.call("compress")
    for k in STATE:
        assert (_canon(k, got[k]) == _canon(k, fe.get(k))).all(), k


# what each stage is expected to leave behind, in terms of the state both sides can serialise
_STAGE_STATE = {
    "init_etc1_images": ["etc1_blocks"],
    "generate_endpoint_clusters": ["endpoint_clusters", "endpoint_parent_clusters"],
    "introduce_new_endpoint_clusters": ["endpoint_clusters"],
    "generate_endpoint_codebook": ["endpoint_cluster_etc_params"],
    "refine_endpoint_clusterization": ["endpoint_clusters"],
    "eliminate_redundant_or_empty_endpoint_clusters": ["endpoint_clusters", "endpoint_cluster_etc_params"],
    "generate_block_endpoint_clusters": ["block_endpoint_clusters_indices"],
    "create_initial_packed_texture": ["encoded_blocks", "orig_encoded_blocks"],
    "generate_selector_clusters": ["selector_cluster_block_indices"],
    "create_optimized_selector_codebook": ["optimized_cluster_selectors"],
    "find_optimal_selector_clusters_for_each_block": ["block_selector_cluster_index", "encoded_blocks", "selector_cluster_block_indices"],
    "introduce_special_selector_clusters": ["optimized_cluster_selectors", "block_selector_cluster_index", "encoded_blocks"],
    "refine_block_endpoints_given_selectors": ["endpoint_cluster_etc_params", "encoded_blocks", "block_endpoint_clusters_indices", "endpoint_clusters"],
    "optimize_selector_codebook": ["optimized_cluster_selectors", "block_selector_cluster_index"],
    "finalize": STATE + ["endpoint_parent_clusters"],
}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("case", ["synth256_l1", "synth256_l4", "synth128_l6", "noise_l4"])
def test_frontend_single_stepped_against_live_reference(hip_ctx, case):
    """Both frontends driven one private stage at a time in the order of basisu_frontend::compress() (frontend.cpp:159-316), the state
    compared after EVERY stage: a difference is pinned to the stage that introduced it, and the lazily materialised forms of our
    clusterings (lists from the per-block maps, parents from the parent-of-vector map) are read in every intermediate state."""
    from basis_universal_amd.etc1s import Etc1sFrontend
    blocks, max_ep, max_sel, level, perceptual = _params(case)
    ref = RefFrontend(blocks, max_ep, max_sel, level, perceptual)
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, level, perceptual)
    trace = []

    def step(stage, arg=0):
        r = ref.call(stage, arg)
        fe.call(stage, arg)
        trace.append(stage)
        for k in _STAGE_STATE.get(stage, []):
            a, b = _canon(k, fe.get(k)), _canon(k, ref.get(k))
            if stage == "generate_selector_clusters":
                # the reference's lists are in TSVQ leaf order here and its next stages only sum over them (frontend.cpp:2267-2330)
                # before find_optimal_selector_clusters_for_each_block rebuilds them ascending; ours are ascending throughout
                a, b = _sorted_lists(a), _sorted_lists(b)
            assert a.shape == b.shape and (a == b).all(), (stage, len(trace), k, a.shape, b.shape)
        return r

    step("init_etc1_images")
    step("init_endpoint_training_vectors")
    step("generate_endpoint_clusters")
    for it in range(ref.call("num_endpoint_codebook_iterations")):
        if it:
            step("introduce_new_endpoint_clusters")
        step("generate_endpoint_codebook", it)
        early_out = False
        if ref.call("endpoint_refinement"):
            early_out = step("refine_endpoint_clusterization") == 0
        step("eliminate_redundant_or_empty_endpoint_clusters")
        if early_out:
            break
    step("generate_block_endpoint_clusters")
    step("create_initial_packed_texture")
    step("generate_selector_clusters")
    if ref.call("use_hierarchical_selector_codebooks"):
        step("compute_selector_clusters_within_each_parent_cluster")
    for it in range(1 if level == 0 else ref.call("num_selector_codebook_iterations")):
        step("create_optimized_selector_codebook", it)
        step("find_optimal_selector_clusters_for_each_block")
        step("introduce_special_selector_clusters")
        if level >= 4 and not step("refine_block_endpoints_given_selectors"):
            break
    step("optimize_selector_codebook")
    step("finalize")
    assert "endpoint_parent_clusters" in _STAGE_STATE["finalize"] and len(trace) >= 12
    fe.close()
    ref.close()


def test_context_close_takes_its_frontends_along():
    """A frontend frees its device buffers through its context; closing the context first (a session fixture torn down before a
    late garbage collection, say) must close the frontend rather than leave it to free through a dead context."""
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import Etc1sFrontend
    ctx = capi.Context()
    fe = Etc1sFrontend(ctx)
    fe.init(to_pixel_blocks(synth(64, 64, 1)), 32, 32, 1, True)
    fe.compress()
    ctx.close()
    assert fe.h is None and ctx.h is None
    fe.close()
    del fe


def test_frontend_outlives_its_context_at_the_c_level():
    """What the Python wrapper arranges for itself must also hold for a C / C++ host (the reference's basis_compressor destroys its accelerator
    context before its frontend member): a context that is destroyed first tells its frontends (bu_hip_on_destroy), they let go of their
    device buffers, their host-side results stay readable, and destroying them afterwards is harmless."""
    import ctypes as C
    from basis_universal_amd import capi, etc1s
    lib = capi.load_library().dll
    F = etc1s.load_frontend_library()
    lib.bu_hip_create_context.restype = C.c_void_p
    lib.bu_hip_destroy_context.argtypes = [C.c_void_p]
    ctx = lib.bu_hip_create_context()
    assert ctx
    blocks = to_pixel_blocks(synth(64, 64, 1))
    F.bu_frontend_create.restype = C.c_void_p
    fe = F.bu_frontend_create()
    assert F.bu_frontend_init(C.c_void_p(fe), C.c_void_p(ctx), blocks.ctypes.data_as(C.c_void_p), None, blocks.shape[0], 32, 32, 1, 1) == 1
    assert F.bu_frontend_compress(C.c_void_p(fe)) == 1
    lib.bu_hip_destroy_context(ctx)
    F.bu_frontend_get.restype = C.c_uint64
    F.bu_frontend_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    assert F.bu_frontend_get(fe, b"encoded_blocks", None, 0) == blocks.shape[0] * 8
    assert F.bu_frontend_compress(C.c_void_p(fe)) == 0          # fails cleanly, no device access through the dead context
    F.bu_frontend_destroy.argtypes = [C.c_void_p]
    F.bu_frontend_destroy(fe)


def test_reinitialised_frontend_serves_the_new_images_tiles(hip_ctx):
    """One frontend object, two device-only images of the same size one after the other: the host copy of the tiles the backend reads
    (get_source_pixel_block) must be the second image's, i.e. the payload must equal a fresh frontend's for that image."""
    from basis_universal_amd.etc1s import Etc1sFrontend
    from basis_universal_amd.backend import Etc1sBackend
    imgs = [synth(128, 96, 41), synth(128, 96, 42)]
    outs = []
    fe = Etc1sFrontend(hip_ctx)
    for reuse in (True, False):
        for i, img in enumerate(imgs):
            if not reuse:
                fe = Etc1sFrontend(hip_ctx)
            blocks = to_pixel_blocks(img)
            d = hip_ctx.upload(blocks)
            fe.init(d, 200, 200, 1, True, n_blocks=blocks.shape[0])
            fe.compress()
            be = Etc1sBackend.from_frontend(fe, [(0, 32, 24)], 1.5, 1.25, 1)
            be.encode()
            outs.append((reuse, i, hashlib.sha256(np.ascontiguousarray(be.get("slice_image_data")).tobytes()).hexdigest()))
            be.close()
            hip_ctx.free(d)
            if not reuse:
                fe.close()
        if reuse:
            fe.close()
    got = {(r, i): h for r, i, h in outs}
    assert got[(True, 0)] == got[(False, 0)] and got[(True, 1)] == got[(False, 1)] and got[(True, 0)] != got[(True, 1)]
