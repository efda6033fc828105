"""-m gpu: bu_frontend_pipeline_* (include/basisu_hip_frontend.h) -- N ETC1S frontends in flight as cooperative tasks on the library's one driver thread. What is
under test: a task that yields wherever its context would block, while other images' kernels are launched from the same thread, ends with exactly the state a
frontend on a context of its own ends with = the reference's (committed golden digests), for every mix of images, levels and lane counts."""
import json
import pathlib

import numpy as np
import pytest

import test_gpu_etc1s_frontend as T

pytestmark = pytest.mark.gpu

MIX = ["synth256_l1", "synth256_l3_flat", "synth256_l4", "synth512_q128", "noise_small_codebooks", "ragged_edges", "synth256_l0", "synth128_l6", "synth256_l2_linear"]


@pytest.mark.parametrize("lanes,drivers", [(1, 1), (3, 1), (8, 1), (6, 2), (8, 4)])
def test_mixed_images_in_flight_equal_the_reference(lanes, drivers):
    from basis_universal_amd.etc1s import FrontendPipeline
    golden = json.loads(T.GOLDEN.read_text())
    pipe = FrontendPipeline(0, lanes, drivers)   # drivers > 1: the lanes dealt out over that many driver threads (bu_frontend_pipeline_create_n)
    tickets = []
    for rep in range(2):
        for case in MIX:
            blocks, max_ep, max_sel, level, perceptual = T._params(case)
            tickets.append((case, pipe.submit(blocks, max_ep, max_sel, level, perceptual)))   # host tiles: uploaded inside the task
    for case, t in reversed(tickets):   # collected in another order than they finish in
        fe = pipe.wait(t)
        got = T._digest({k: fe.get(k) for k in T.STATE})
        fe.close()
        assert got == golden[case]["digests"], (case, lanes, [k for k in got if got[k] != golden[case]["digests"][k]])
    st = pipe.stats()
    assert st["jobs"] == len(tickets) and st["yields"] > len(tickets)
    pipe.close()


def test_resident_tiles_backend_and_release(hip_ctx):
    """tiles already in HBM (device pointer), the host backend on a pipelined frontend (its device call-backs run on the context the pipeline lent it),
    contexts handed back on release and reused by later jobs"""
    import hashlib
    from basis_universal_amd.etc1s import FrontendPipeline
    from basis_universal_amd.backend import Etc1sBackend
    gb = json.loads((pathlib.Path(__file__).parent / "golden" / "etc1s_backend_digests.json").read_text())["synth256_l1"]
    blocks, max_ep, max_sel, level, perceptual = T._params("synth256_l1")
    d = hip_ctx.upload(blocks)
    pipe = FrontendPipeline(0, 2)
    golden = json.loads(T.GOLDEN.read_text())["synth256_l1"]["digests"]
    for rnd in range(3):
        ts = [pipe.submit(d, max_ep, max_sel, level, perceptual, n_blocks=blocks.shape[0]) for _ in range(4)]
        assert all(isinstance(t, int) and t > 0 for t in ts)
        fes = [pipe.wait(t) for t in ts]
        for fe in fes:
            assert T._digest({k: fe.get(k) for k in T.STATE}) == golden
        be = Etc1sBackend.from_frontend(fes[0], [tuple(s) for s in gb["slices"]], 1.5, 1.25, 1)
        assert be.encode() == gb["compressed_bytes"]
        assert {k: hashlib.sha256(np.ascontiguousarray(be.get(k)).tobytes()).hexdigest() for k in gb["digests"]} == gb["digests"]
        be.close()
        for fe in fes:
            fe.close()
    pipe.close()
    hip_ctx.free(d)


def test_a_failing_job_fails_alone():
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import FrontendPipeline
    golden = json.loads(T.GOLDEN.read_text())["synth256_l1"]["digests"]
    blocks, max_ep, max_sel, level, perceptual = T._params("synth256_l1")
    pipe = FrontendPipeline(0, 2)
    good = pipe.submit(blocks, max_ep, max_sel, level, perceptual)
    bad = pipe.submit(blocks, 0, max_sel, level, perceptual)          # bad max_endpoint_clusters: init refuses
    good2 = pipe.submit(blocks, max_ep, max_sel, level, perceptual)
    with pytest.raises(capi.HipError, match="max_endpoint_clusters"):
        pipe.wait(bad)
    for t in (good, good2):
        fe = pipe.wait(t)
        assert T._digest({k: fe.get(k) for k in T.STATE}) == golden
        fe.close()
    with pytest.raises(capi.HipError):
        pipe.wait(bad)   # the ticket is gone
    pipe.close()


def test_headline_image_four_in_flight(hip_ctx):
    """BASELINE configs[1] (4096^2 q128 level 1), eight images through four lanes, single-threaded and the reference's 8-thread codebook configuration mixed:
    every one = the reference's digests"""
    from helpers import synth, to_pixel_blocks
    from basis_universal_amd.etc1s import FrontendPipeline
    big = json.loads((pathlib.Path(__file__).parent / "golden" / "etc1s_big_digests.json").read_text())
    g1, g8 = big["synth4096_q128"], big["synth4096_q128_t8"]
    blocks = to_pixel_blocks(synth(4096, 4096, 1234))
    n = blocks.shape[0]
    d = hip_ctx.upload(blocks)   # (not a torch tensor: torch brings a HIP runtime of its own, which cannot start once this process has initialised the system's)
    pipe = FrontendPipeline(0, 4)
    ts = [(pipe.submit(d, g1["max_endpoint_clusters"], g1["max_selector_clusters"], g1["level"], g1["perceptual"], n_blocks=n, max_threads=(8 if i % 4 == 3 else 0)),
           g8 if i % 4 == 3 else g1) for i in range(8)]
    for t, g in ts:
        fe = pipe.wait(t)
        got = T._digest({k: fe.get(k) for k in g["frontend_digests"]})
        fe.close()
        assert got == g["frontend_digests"], [k for k in got if got[k] != g["frontend_digests"][k]]
    pipe.close()
    hip_ctx.free(d)
