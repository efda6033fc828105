"""-m gpu: BASELINE.json's own configurations at FULL size against the reference, through committed goldens (tools/gen_golden_big.py ran
the real reference -- oracle/_ref, single-threaded and with job pools of 2 / 4 / 8 threads (the tool's default configuration) -- in the build container; the GPU box has no
/root/reference and could not afford its minutes anyway):

  configs[0]  kodim03.png 768x512 ETC1S -q128            pixels + the reference TOOL's .basis / .ktx2 bytes in tests/golden/kodim03.npz
  configs[1]  4096x4096 synthetic RGBA (seed 1234) -q128  frontend state digests + backend payload digests
  configs[2]  4096x4096 synthetic RGBA (seed 1234) UASTC   ALL 1,048,576 blocks at level 2: sha256 of the whole output + per 65,536-block chunk
  configs[3]  8192x8192 synthetic RGBA (seed 5678) -q255  the same, at the 8192 / 16128 cluster codebooks no smaller test reaches
  configs[4]  the Kodak batch, UASTC + RDO                 tests/test_gpu_kodak24.py

Everything compared is integer data; the comparison is exact."""
import hashlib
import json
import pathlib

import numpy as np
import pytest

from helpers import synth, to_pixel_blocks, uniform_random, kodak_mosaic, endpoint_cube
import test_gpu_etc1s_frontend as T

pytestmark = pytest.mark.gpu
GOLDEN = json.loads((pathlib.Path(__file__).parent / "golden" / "etc1s_big_digests.json").read_text())
KODIM03 = pathlib.Path(__file__).parent / "golden" / "kodim03.npz"


def _frontend_and_backend(hip_ctx, img, g):
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend
    blocks = to_pixel_blocks(img)
    assert quality_to_clusters(g["quality"], blocks.shape[0]) == (g["max_endpoint_clusters"], g["max_selector_clusters"])
    fe = Etc1sFrontend(hip_ctx, max_threads=g.get("threads", 1))
    fe.init(blocks, g["max_endpoint_clusters"], g["max_selector_clusters"], g["level"], g["perceptual"])
    fe.compress()
    got = T._digest({k: fe.get(k) for k in T.STATE})
    assert got == g["frontend_digests"], {k: (got[k][:12], g["frontend_digests"][k][:12]) for k in got if got[k] != g["frontend_digests"][k]}
    b = g["backend"]
    be = Etc1sBackend.from_frontend(fe, [tuple(s) for s in b["slices"]], b["thresholds"][0], b["thresholds"][1], g["level"])
    assert be.encode() == b["compressed_bytes"]
    dg = {k: hashlib.sha256(np.ascontiguousarray(be.get(k)).tobytes()).hexdigest() for k in b["digests"]}
    assert dg == b["digests"]
    return fe, be


def test_config1_synth4096_q128(hip_ctx):
    fe, be = _frontend_and_backend(hip_ctx, synth(4096, 4096, 1234), GOLDEN["synth4096_q128"])
    be.close(); fe.close()


def test_config3_synth8192_q255(hip_ctx):
    g = GOLDEN["synth8192_q255"]
    assert (g["max_endpoint_clusters"], g["max_selector_clusters"]) == (8192, 16128)   # comp.cpp:3325-3379
    fe, be = _frontend_and_backend(hip_ctx, synth(8192, 8192, 5678), g)
    be.close(); fe.close()


@pytest.mark.parametrize("threads", [2, 4, 8])
def test_config1_synth4096_q128_multithreaded_reference(hip_ctx, threads):
    """The reference's DEFAULT configuration: a job pool of T threads -> T-way partitioned selector codebook (674,691 distinct selector vectors >= 262,144;
    generate_hierarchical_codebook_threaded_internal, enc.h:2086-2215). Goldens: tools/gen_golden_big.py with RefFrontend(threads=T)."""
    g = GOLDEN[f"synth4096_q128_t{threads}"]
    assert g["threads"] == threads and g["frontend_digests"] != GOLDEN["synth4096_q128"]["frontend_digests"]
    fe, be = _frontend_and_backend(hip_ctx, synth(4096, 4096, 1234), g)
    be.close(); fe.close()


@pytest.mark.parametrize("threads", [2, 4, 8])
def test_config3_synth8192_q255_multithreaded_reference(hip_ctx, threads):
    g = GOLDEN[f"synth8192_q255_t{threads}"]
    assert g["threads"] == threads and g["frontend_digests"] != GOLDEN["synth8192_q255"]["frontend_digests"]
    fe, be = _frontend_and_backend(hip_ctx, synth(8192, 8192, 5678), g)
    be.close(); fe.close()


@pytest.mark.parametrize("case,single", [("synth3072_q200_l2_t8", "synth3072_q200_l2"), ("synth3072_q90_l4_t4", "synth3072_q90_l4")])
def test_other_levels_in_both_thread_configurations(hip_ctx, case, single):
    """The reference's two thread configurations at compression levels 2 (the library default: other parent codebook sizes, the backend's call back into the frontend)
    and 4 (new-cluster insertion, endpoint refinement given selectors), 3072^2 (past the 262,144-vector gate of the partitioned codebook build): frontend state and
    backend payloads of the reference run with a pool of 8 / 4 threads and of the single-threaded run (tools/gen_golden_big.py)."""
    g, g1 = GOLDEN[case], GOLDEN[single]
    assert g["threads"] > 1 and g1["threads"] == 1 and g["level"] == g1["level"] and g["frontend_digests"] != g1["frontend_digests"]
    for gg in (g, g1):
        fe, be = _frontend_and_backend(hip_ctx, synth(3072, 3072, 4321), gg)
        be.close(); fe.close()


OTHER_DISTRIBUTIONS = {
    # SURVEY 8d's second distribution at FULL size -- uniform-random RGB, seed 42: one distinct selector vector per block (1,048,431 / ~4.19 M: the largest trees, the
    # largest T-way partitions and refine lists the builders can meet), photographic statistics at full size (the reference's 24 Kodak test images as one mosaic), and the
    # worst case for the ENDPOINT builder (nearly every ETC1S endpoint occurs: ~2 x 10^5 distinct 6-float vectors in ONE tree -- the ceiling is 236,235, below the
    # reference's 262,144 gate, tests/test_host_logic.py -- so the 6-float many-workgroup passes meet nodes of 10^5 members in a real run)
    "noise4096_q128": lambda: uniform_random(4096, 4096, 42),
    "kodak4096_q128": lambda: kodak_mosaic(4096, 4096),
    "cube4096_q128": lambda: endpoint_cube(4096, 4096, 7),
    "noise8192_q255": lambda: uniform_random(8192, 8192, 42),
}


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("name", sorted(OTHER_DISTRIBUTIONS))
def test_other_distributions_at_full_size_in_both_thread_configurations(hip_ctx, name, threads):
    case = name if threads == 1 else f"{name}_t{threads}"
    if case not in GOLDEN:
        pytest.skip(f"no golden for {case} (tools/gen_golden_big.py {case})")
    g = GOLDEN[case]
    assert g.get("threads", 1) == threads
    if threads > 1:
        assert g["distinct_vectors"]["selector"] >= 262144 and g["frontend_digests"] != GOLDEN[name]["frontend_digests"]   # past the gate: a different codebook
    fe, be = _frontend_and_backend(hip_ctx, OTHER_DISTRIBUTIONS[name](), g)
    be.close(); fe.close()


def test_config0_kodim03_q128_file_equals_the_reference_tools(hip_ctx):
    """the row of the reference's own golden table (basisu_tool.cpp:6751), held to the bytes instead of its 4.5 % / 0.3 dB tolerance"""
    from helpers import basis_file_key_values, ktx2_file_key_values
    z = np.load(KODIM03)
    img = np.ascontiguousarray(z["rgba"])
    assert img.shape == (512, 768, 4)
    fe, be = _frontend_and_backend(hip_ctx, img, GOLDEN["kodim03_q128"])
    mine = be.basis_file(key_values=basis_file_key_values(z["tool_basis"]))
    assert mine.shape == z["tool_basis"].shape and (mine == z["tool_basis"]).all()
    mine2 = be.ktx2_file(key_values=ktx2_file_key_values(z["tool_ktx2"]))
    assert mine2.shape == z["tool_ktx2"].shape and (mine2 == z["tool_ktx2"]).all()
    be.close(); fe.close()


def test_config2_synth4096_uastc_level2_every_block(hip_ctx):
    """BASELINE configs[2] at full size against the reference's output for EVERY block (tools/gen_golden_uastc_big.py: ~80 s of oracle/_ref on one core):
    a whole-output digest, with per-chunk digests so that a failure names its neighbourhood."""
    from basis_universal_amd import uastc
    g = json.loads((pathlib.Path(__file__).parent / "golden" / "uastc_big_digests.json").read_text())["synth4096_l2"]
    blocks = to_pixel_blocks(synth(g["width"], g["height"], g["seed"]))
    assert blocks.shape[0] == g["n_blocks"] == 1048576
    got = uastc.encode_uastc_blocks(hip_ctx, blocks, g["flags"])
    c = g["chunk_blocks"]
    bad = [i for i, want in enumerate(g["chunk_sha256"]) if hashlib.sha256(got[i * c:(i + 1) * c].tobytes()).hexdigest() != want]
    assert not bad, f"chunks {bad} of {len(g['chunk_sha256'])} differ from the reference"
    assert hashlib.sha256(got.tobytes()).hexdigest() == g["sha256"]
    assert np.bincount(got[:, 0] & 0x7F, minlength=128).tolist() == g["mode_histogram"]
