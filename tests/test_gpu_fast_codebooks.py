"""-m gpu: SURVEY.md 8f row f3, the codebook builders' FAST mode (k-means with the assignment GEMM on the matrix cores instead of the TSVQ,
basis_universal_amd/csrc/kmeans_kernels.hip). It is explicitly NOT bit-identical to the reference, so it is held to the reference's OWN
acceptance test instead (basisu_tool.cpp:6786-6793: .basis size within 4.5 %, RGB(A) PSNR not more than 0.3 dB lower) -- against the
bit-exact path of this repository, which equals the reference's output -- and to run-to-run determinism."""
import hashlib
import pathlib

import numpy as np
import pytest

from helpers import synth, uniform_random, to_pixel_blocks, decode_etc1s_blocks, psnr

pytestmark = pytest.mark.gpu
KODIM03 = pathlib.Path(__file__).parent / "golden" / "kodim03.npz"

SIZE_TOL = 0.045   # basisu_tool.cpp:6793
PSNR_TOL = 0.3     # basisu_tool.cpp:6786-6792


def _encode(hip_ctx, img, quality, fast):
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend, default_params
    h, w = img.shape[:2]
    nbx, nby = (w + 3) // 4, (h + 3) // 4
    blocks = to_pixel_blocks(img)
    max_ep, max_sel = quality_to_clusters(quality, blocks.shape[0])
    fe = Etc1sFrontend(hip_ctx, fast_codebooks=fast)
    fe.init(blocks, max_ep, max_sel, 1, True)
    fe.compress()
    enc = fe.get("encoded_blocks").reshape(-1, 8).copy()
    ept, selt = default_params(quality, 1)
    be = Etc1sBackend.from_frontend(fe, [(0, nbx, nby, w, h, 0, 0, 0)], ept, selt, 1)
    be.encode()
    size = len(be.basis_file())
    clusters = (int(fe.get("endpoint_clusters", np.uint32)[0]), int(fe.get("selector_cluster_block_indices", np.uint32)[0]))
    be.close(); fe.close()
    dec = decode_etc1s_blocks(enc, nbx, nby)[:h, :w]
    return {"size": size, "psnr": psnr(dec, img[..., :3]), "clusters": clusters, "digest": hashlib.sha256(enc.tobytes()).hexdigest()}


def _kodim03():
    return np.ascontiguousarray(np.load(KODIM03)["rgba"])


CASES = {
    "kodim03_q128": (_kodim03, 128),
    "kodim03_q64": (_kodim03, 64),
    "synth1024_q128": (lambda: synth(1024, 1024, 1234), 128),
    "synth512_q200": (lambda: synth(512, 512, 99), 200),
    "noise256_q128": (lambda: uniform_random(256, 256, 42), 128),
    "ragged_q128": (lambda: synth(260, 132, 3)[:131, :258], 128),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_fast_codebooks_within_the_references_tolerances(hip_ctx, case):
    img_fn, quality = CASES[case]
    img = img_fn()
    exact = _encode(hip_ctx, img, quality, fast=False)
    fast = _encode(hip_ctx, img, quality, fast=True)
    again = _encode(hip_ctx, img, quality, fast=True)
    assert fast["digest"] == again["digest"], "the fast mode must be deterministic"
    assert fast["digest"] != exact["digest"] or case.startswith("noise"), "fast mode produced the bit-exact path's output: is it wired up?"
    assert fast["psnr"] >= exact["psnr"] - PSNR_TOL, (fast, exact)
    assert fast["size"] <= exact["size"] * (1 + SIZE_TOL), (fast, exact)
    print(f"{case}: exact {exact['size']} B {exact['psnr']:.3f} dB {exact['clusters']} | fast {fast['size']} B {fast['psnr']:.3f} dB {fast['clusters']}")


def test_fast_codebooks_full_size_image(hip_ctx):
    """BASELINE's 4096^2 image: the gate at the size the bench runs at (device-resident tiles)"""
    img = synth(4096, 4096, 1234)
    exact = _encode(hip_ctx, img, 128, fast=False)
    fast = _encode(hip_ctx, img, 128, fast=True)
    assert fast["psnr"] >= exact["psnr"] - PSNR_TOL and fast["size"] <= exact["size"] * (1 + SIZE_TOL), (fast, exact)
    print(f"synth4096: exact {exact} | fast {fast}")
