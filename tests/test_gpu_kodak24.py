"""-m gpu: the reference's own test images, kodim01..24 (tests/golden/kodak24.npz), through both codecs on the MI355X.

  * BASELINE.json configs[4]: the 24 images as ONE resident batch through encode_uastc level 2 + uastc_rdo (lambda 1.0) -- the form bench.py times --
    held to the real reference's output image by image (sha256 of every image's blocks, before and after RDO, for the tool's two strip
    counts: 1 = -no_multithreading, 4 = min(4, threads) of comp.cpp:2078). BASELINE asks for a PSNR gate; the comparison here is exact.
  * The rows of the reference's own golden table (basisu_tool.cpp:6737-6776, `basisu -test`): ETC1S -q 128 and UASTC on every Kodak image,
    first against the reference's exact bytes (frontend state, backend payloads, the .basis file the tool writes), then -- like the
    reference's test_mode_ldr does -- against the table's file sizes and RGBA PSNRs within its own tolerances (4.5 % / 0.3 dB).

Goldens: tools/gen_golden_kodak.py ran oracle/_ref (single-threaded, the pinned configuration) in the build container."""
import hashlib
import json
import pathlib

import numpy as np
import pytest

import helpers
import test_gpu_etc1s_frontend as T
from basis_universal_amd import uastc

pytestmark = pytest.mark.gpu
HERE = pathlib.Path(__file__).resolve().parent
GOLDEN = json.loads((HERE / "golden" / "kodak24_digests.json").read_text())
NAMES = sorted(GOLDEN["images"])
ETC1S_FILESIZE_THRESHOLD, PSNR_THRESHOLD = 0.045, 0.3   # basisu_tool.cpp:6786-6793


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def kodak():
    z = np.load(HERE / "golden" / "kodak24.npz")
    imgs = {}
    for name in NAMES:
        rgb = z[name]
        assert sha(rgb) == GOLDEN["images"][name]["rgb_sha256"]
        imgs[name] = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
    return imgs


@pytest.fixture(scope="module")
def batch(kodak):
    """all 24 images as one block array + the block range of every image"""
    parts = [helpers.to_pixel_blocks(kodak[n]) for n in NAMES]
    ofs = np.cumsum([0] + [p.shape[0] for p in parts])
    return np.ascontiguousarray(np.concatenate(parts)), ofs


def test_config4_uastc_level2_every_image_equals_the_reference(hip_ctx, batch):
    blocks, ofs = batch
    assert blocks.shape[0] == 24 * 24576
    got = uastc.encode_uastc_blocks(hip_ctx, blocks, 2)
    for i, name in enumerate(NAMES):
        assert sha(got[ofs[i]:ofs[i + 1]]) == GOLDEN["images"][name]["uastc_l2"], name


@pytest.mark.parametrize("jobs", [1, 4])
def test_config4_uastc_rdo_lambda1_every_image_equals_the_reference(hip_ctx, batch, kodak, jobs):
    """One launch over the whole batch: 24 x jobs strips of 24576 / jobs blocks walk concurrently (the bench's configuration), and every image ends
    with the bytes the reference's serial walk leaves."""
    blocks, ofs = batch
    packed = uastc.encode_uastc_blocks(hip_ctx, blocks, 2)
    got, info = uastc.uastc_rdo(hip_ctx, packed, blocks, uastc.RdoParams(m_lambda=1.0), 2, 24 * jobs)
    assert info["strips"] == 24 * jobs
    total_modified = 0
    for i, name in enumerate(NAMES):
        g = GOLDEN["images"][name]
        mine = got[ofs[i]:ofs[i + 1]]
        assert sha(mine) == g[f"uastc_l2_rdo1_jobs{jobs}"], name
        total_modified += g[f"uastc_rdo1_jobs{jobs}_modified"]
        # the PSNR gate BASELINE names, on top of the exact comparison: our decode of our blocks vs the reference's decode of its own
        h, w = kodak[name].shape[:2]
        p = helpers.psnr(helpers.host_decode_uastc(mine, w // 4, h // 4), kodak[name])
        assert abs(p - g[f"uastc_psnr_rgba_rdo1_jobs{jobs}"]) < 1e-3, (name, p)
    assert int((got != packed).any(axis=1).sum()) == total_modified


@pytest.mark.parametrize("lanes", [1, 3])
def test_config4_pipeline_images_in_flight_equal_the_reference(hip_ctx, batch, kodak, lanes):
    """BASELINE configs[4] through the library's pipeline (bu_hip_uastc_pipeline_*): the 24 images as 24 separate submissions (encode_uastc level 2 + uastc_rdo lambda 1.0,
    4 strips each = the tool's min(4, threads)), `lanes` of them in flight on private streams with no host synchronisation in between, then the whole batch twice more as
    single submissions with 24 and 96 strips while the tail of the per-image ones is still running. Every image ends with the reference's bytes."""
    blocks, ofs = batch
    n = blocks.shape[0]
    d_px = hip_ctx.upload(blocks)
    d_img = hip_ctx.alloc(n * 16)
    d_b1, d_b4 = hip_ctx.alloc(n * 16), hip_ctx.alloc(n * 16)
    params = uastc.RdoParams(m_lambda=1.0)
    pipe = uastc.UastcPipeline(hip_ctx, lanes, n, 2, 96)
    try:
        if lanes == 1:   # one lane: a ticket's statistics are there until the lane is used again -- ask for them before the next submission
            stats = []
            for i in range(len(NAMES)):
                t = pipe.submit(d_px + int(ofs[i]) * 64, int(ofs[i + 1] - ofs[i]), d_img + int(ofs[i]) * 16, params, 2, 4)
                stats.append(pipe.wait(t))
        else:
            tickets = [pipe.submit(d_px + int(ofs[i]) * 64, int(ofs[i + 1] - ofs[i]), d_img + int(ofs[i]) * 16, params, 2, 4) for i in range(len(NAMES))]
        t1 = pipe.submit(d_px, n, d_b1, params, 2, 24)
        s1 = pipe.wait(t1) if lanes == 1 else None
        t4 = pipe.submit(d_px, n, d_b4, params, 2, 96)
        s4 = pipe.wait(t4)
        if lanes > 1:
            s1 = pipe.wait(t1)
        pipe.wait(0)
    finally:
        pipe.close()
    per_image = hip_ctx.download(d_img, (n, 16), np.uint8)
    b1, b4 = hip_ctx.download(d_b1, (n, 16), np.uint8), hip_ctx.download(d_b4, (n, 16), np.uint8)
    for d in (d_px, d_img, d_b1, d_b4):
        hip_ctx.free(d)
    for i, name in enumerate(NAMES):
        g = GOLDEN["images"][name]
        assert sha(per_image[ofs[i]:ofs[i + 1]]) == g["uastc_l2_rdo1_jobs4"], name
        assert sha(b4[ofs[i]:ofs[i + 1]]) == g["uastc_l2_rdo1_jobs4"], name
        assert sha(b1[ofs[i]:ofs[i + 1]]) == g["uastc_l2_rdo1_jobs1"], name
        if lanes == 1:
            # (the walk's counter includes the rare block it rewrote with the bytes it already had; the golden counts changed bytes)
            assert 0 <= stats[i]["modified"] - g["uastc_rdo1_jobs4_modified"] <= 4 and stats[i]["strips"] == 4, (name, stats[i])
    assert s4["strips"] == 96 and s1["strips"] == 24
    assert 0 <= s4["modified"] - sum(GOLDEN["images"][nm]["uastc_rdo1_jobs4_modified"] for nm in NAMES) <= 4 * len(NAMES)


def test_reference_table_uastc(hip_ctx, batch, kodak):
    """basisu -test, UASTC rows: basis_compress() with no level bits = pack level 0; RGBA PSNR within 0.3 dB of the table. Exact first."""
    blocks, ofs = batch
    got = uastc.encode_uastc_blocks(hip_ctx, blocks, 0)
    for i, name in enumerate(NAMES):
        g = GOLDEN["images"][name]
        mine = got[ofs[i]:ofs[i + 1]]
        assert sha(mine) == g["uastc_l0"], name
        h, w = kodak[name].shape[:2]
        p = helpers.psnr(helpers.host_decode_uastc(mine, w // 4, h // 4), kodak[name])
        assert abs(p - GOLDEN["reference_table"][name]["uastc_psnr"]) <= PSNR_THRESHOLD, (name, p)


def _etc1s(hip_ctx, img, g):
    """frontend + backend of one image in configuration g -> (frontend, backend, .basis bytes), every stage held to the reference's digests"""
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend
    blocks = helpers.to_pixel_blocks(img)
    assert quality_to_clusters(g["quality"], blocks.shape[0]) == (g["max_endpoint_clusters"], g["max_selector_clusters"])
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, g["max_endpoint_clusters"], g["max_selector_clusters"], g["level"], g["perceptual"])
    fe.compress()
    b = g["backend"]
    be = Etc1sBackend.from_frontend(fe, [tuple(s) for s in b["slices"]], b["thresholds"][0], b["thresholds"][1], g["level"])
    assert be.encode() == b["compressed_bytes"]
    assert {k: sha(be.get(k)) for k in b["digests"]} == b["digests"]
    got = T._digest({k: fe.get(k) for k in T.STATE})   # after the backend: above level 1 it re-optimises the endpoint codebook through the frontend
    assert got == g["frontend_digests_after_backend"], [k for k in got if got[k] != g["frontend_digests_after_backend"][k]]
    data = be.basis_file(key_values=[(k, bytes.fromhex(v)) for k, v in g["tool_basis_key_values"]])
    assert data.size == g["tool_basis_size"] and sha(data) == g["tool_basis_sha256"]
    return fe, be, data


def _psnr_rgba(fe, be, img):
    h, w = img.shape[:2]
    decoded = helpers.decode_backend_output(fe, be, w // 4, h // 4)
    return helpers.psnr(np.concatenate([decoded, np.full((h, w, 1), 255, np.uint8)], axis=2), img)


@pytest.mark.parametrize("name", NAMES)
def test_etc1s_q128_command_line_defaults(hip_ctx, kodak, name):
    """`basisu -etc1s -q 128` (comp level 1, sRGB metrics; BASELINE configs[0] is the kodim03 row of this): frontend state, backend payloads and the
    written .basis file = the reference tool's, byte for byte."""
    g = GOLDEN["images"][name]["etc1s_q128"]
    fe, be, _ = _etc1s(hip_ctx, kodak[name], g)
    assert abs(_psnr_rgba(fe, be, kodak[name]) - g["psnr_rgba"]) < 1e-3
    be.close(); fe.close()


@pytest.mark.parametrize("name", NAMES)
def test_reference_table_etc1s_q128(hip_ctx, kodak, name):
    """basisu -test, ETC1S quality-128 rows = basis_compress() with the library defaults (comp level 2, LINEAR metrics): exact against the reference
    built here first, then -- the reference's own acceptance rule -- size and RGBA PSNR within 4.5 % / 0.3 dB of its table."""
    g = GOLDEN["images"][name]["etc1s_q128_table"]
    assert (g["level"], g["perceptual"]) == (2, False)
    fe, be, data = _etc1s(hip_ctx, kodak[name], g)
    row = GOLDEN["reference_table"][name]
    assert abs(data.size / row["etc1s_q128_size"] - 1.0) <= ETC1S_FILESIZE_THRESHOLD, data.size
    p = _psnr_rgba(fe, be, kodak[name])
    assert abs(p - g["psnr_rgba"]) < 1e-3 and abs(p - row["etc1s_q128_psnr"]) <= PSNR_THRESHOLD, p
    be.close(); fe.close()
