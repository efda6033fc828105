"""-m gpu: the whole ETC1S encoder path -- resident frontend (HIP kernels) followed by the host backend (bu::etc1s_backend) -- against the
real reference frontend + basisu_backend::encode() (oracle/_ref): every output byte and the per-block state. At compression levels above 1
the backend calls back into the frontend (reoptimize_remapped_endpoints -> the forced-selector cluster fit on the device), so the frontend
state AFTER the backend is compared as well."""
import numpy as np
import pytest

from helpers import have_ref, RefFrontend, synth, uniform_random, to_pixel_blocks

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")

OUTPUTS = ["endpoint_palette", "selector_palette", "slice_image_tables", "slice_image_crcs", "num_endpoints", "num_selectors",
           "encoder_blocks", "endpoint_remap_old_to_new", "selector_remap_new_to_old"]
FRONTEND_AFTER = ["endpoint_cluster_etc_params", "block_endpoint_clusters_indices", "encoded_blocks"]

CASES = {
    # name: (image, max_ep, max_sel, level, perceptual, slices, endpoint thresh, selector thresh)
    "l1": (lambda: synth(256, 192, 1234), 400, 500, 1, True, [(0, 64, 48)], 1.5, 1.25),
    "l2": (lambda: synth(256, 192, 11), 300, 300, 2, True, [(0, 64, 48)], 1.5, 1.25),
    "l2_linear_two_slices": (lambda: synth(256, 192, 12), 300, 400, 2, False, [(0, 64, 32), (2048, 64, 16)], 1.5, 1.25),
    "l3_strong_rdo": (lambda: synth(192, 128, 13), 200, 256, 3, True, [(0, 48, 32)], 3.0, 2.0),
    "l4": (lambda: synth(192, 128, 14), 200, 200, 4, True, [(0, 48, 32)], 1.5, 1.25),
    "l6": (lambda: synth(128, 128, 15), 128, 160, 6, True, [(0, 32, 32)], 1.5, 1.25),
    "noise_l2": (lambda: uniform_random(96, 64, 7), 100, 100, 2, True, [(0, 24, 16)], 2.0, 1.5),
    "l2_no_rdo": (lambda: synth(128, 128, 16), 128, 128, 2, True, [(0, 32, 32)], 0.0, 0.0),
}


def _canon(name, a):
    if name == "endpoint_cluster_etc_params":
        a = a.reshape(-1, 16).copy(); a[:, 4:8] = 0
    return a


@needs_ref
@pytest.mark.parametrize("case", sorted(CASES))
def test_frontend_plus_backend_matches_reference(hip_ctx, case):
    from basis_universal_amd.etc1s import Etc1sFrontend
    from basis_universal_amd.backend import Etc1sBackend
    img_fn, max_ep, max_sel, level, perceptual, slices, ept, selt = CASES[case]
    blocks = to_pixel_blocks(img_fn())
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, level, perceptual)
    fe.compress()
    be = Etc1sBackend.from_frontend(fe, slices, ept, selt, level)
    total = be.encode()
    ref = RefFrontend(blocks, max_ep, max_sel, level, perceptual)
    ref.call("compress")
    ref_total, _ = ref.backend_run(slices, ept, selt)
    assert total == ref_total
    for k in OUTPUTS:
        a, b = be.get(k), ref.backend_get(k)
        assert a.shape == b.shape and (a == b).all(), k
    for s in range(len(slices)):
        a, b = be.get("slice_image_data", s), ref.backend_get("slice_image_data", s)
        assert a.shape == b.shape and (a == b).all(), ("slice_image_data", s)
    for k in FRONTEND_AFTER:
        a, b = _canon(k, fe.get(k)), _canon(k, ref.get(k))
        assert a.shape == b.shape and (a == b).all(), ("frontend after backend", k)
    be.close(); fe.close(); ref.close()


def _golden_cases():
    import json, pathlib
    return json.loads((pathlib.Path(__file__).parent / "golden" / "etc1s_backend_digests.json").read_text())


@pytest.mark.parametrize("case", sorted(_golden_cases()))
def test_frontend_plus_backend_matches_golden(hip_ctx, case):
    """The committed digests of the reference's payloads (tools/gen_golden_backend.py): needs no reference build at run time."""
    import hashlib
    import test_gpu_etc1s_frontend as T
    from basis_universal_amd.etc1s import Etc1sFrontend
    from basis_universal_amd.backend import Etc1sBackend
    g = _golden_cases()[case]
    blocks, max_ep, max_sel, level, perceptual = T._params(case)
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, level, perceptual)
    fe.compress()
    be = Etc1sBackend.from_frontend(fe, [tuple(s) for s in g["slices"]], 1.5, 1.25, level)
    assert be.encode() == g["compressed_bytes"]
    got = {k: hashlib.sha256(np.ascontiguousarray(be.get(k)).tobytes()).hexdigest() for k in g["digests"]}
    assert got == g["digests"], [k for k in got if got[k] != g["digests"][k]]
    be.close(); fe.close()


@needs_ref
def test_backend_on_device_only_tiles(hip_ctx):
    """The frontend was given tiles that live in HBM only: the backend fetches its host copy through the frontend."""
    from basis_universal_amd.etc1s import Etc1sFrontend
    from basis_universal_amd.backend import Etc1sBackend
    blocks = to_pixel_blocks(synth(128, 96, 5))
    d = hip_ctx.upload(blocks)
    fe = Etc1sFrontend(hip_ctx)
    fe.init(d, 100, 100, 1, True, n_blocks=blocks.shape[0])
    fe.compress()
    be = Etc1sBackend.from_frontend(fe, [(0, 32, 24)], 1.5, 1.25, 1)
    total = be.encode()
    ref = RefFrontend(blocks, 100, 100, 1, True)
    ref.call("compress")
    assert total == ref.backend_run([(0, 32, 24)])[0]
    assert (be.get("slice_image_data") == ref.backend_get("slice_image_data")).all()
    be.close(); fe.close(); hip_ctx.free(d); ref.close()


@pytest.mark.skipif(not __import__("helpers").have_ref_cli(), reason="oracle/_ref/basisu not present")
@pytest.mark.parametrize("w,h,seed,quality,level", [(256, 192, 1234, 128, 1), (130, 67, 3, 200, 1), (192, 128, 5, 128, 2), (128, 128, 6, 90, 4)])
def test_whole_encoder_matches_reference_command_line(hip_ctx, tmp_path, w, h, seed, quality, level):
    """Drop-in, end to end: the file the reference TOOL writes for a PNG (`basisu -basis -etc1s -q N -comp_level L`) against tiles ->
    resident frontend (HIP) -> host backend -> container writer, byte for byte."""
    from helpers import save_png, run_ref_cli, basis_file_key_values
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend, default_params
    img = np.ascontiguousarray(synth((w + 3) // 4 * 4, (h + 3) // 4 * 4, seed)[:h, :w])
    save_png(tmp_path / "x.png", img)
    cli = run_ref_cli(tmp_path / "x.png", "-etc1s", "-q", str(quality), "-comp_level", str(level))
    blocks = to_pixel_blocks(img)
    max_ep, max_sel = quality_to_clusters(quality, blocks.shape[0])
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, level, True)
    fe.compress()
    ept, selt = default_params(quality, level)
    be = Etc1sBackend.from_frontend(fe, [(0, (w + 3) // 4, (h + 3) // 4, w, h, 0, 0, 0)], ept, selt, level)
    be.encode()
    mine = be.basis_file(key_values=basis_file_key_values(cli))
    assert mine.shape == cli.shape and (mine == cli).all()
    from helpers import ktx2_file_key_values
    cli2 = run_ref_cli(tmp_path / "x.png", "-etc1s", "-q", str(quality), "-comp_level", str(level), ktx2=True)   # the tool's default container
    mine2 = be.ktx2_file(key_values=ktx2_file_key_values(cli2))
    assert mine2.shape == cli2.shape and (mine2 == cli2).all()
    be.close(); fe.close()


@pytest.mark.skipif(not __import__("helpers").have_ref_cli(), reason="oracle/_ref/basisu not present")
@pytest.mark.parametrize("level,rdo,alpha", [(2, None, False), (1, None, True), (2, 1.0, False), (3, 2.0, True)])
def test_uastc_whole_encoder_matches_reference_command_line(hip_ctx, tmp_path, level, rdo, alpha):
    """`basisu -basis -uastc -uastc_level L [-uastc_rdo_l X]` for a PNG against tiles -> HIP encode_uastc (-> HIP uastc_rdo) -> container
    writer, byte for byte."""
    from helpers import save_png, run_ref_cli, basis_file_key_values
    from basis_universal_amd import uastc
    from basis_universal_amd.backend import uastc_basis_file
    w, h = 192, 128
    img = synth(w, h, 60 + level)
    if alpha:
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 3] = np.clip(140 + 110 * np.sin(xx / 19.0 + yy / 31.0), 0, 255).astype(np.uint8)
    save_png(tmp_path / "x.png", img)
    cli = run_ref_cli(tmp_path / "x.png", "-uastc", "-uastc_level", str(level), *(["-uastc_rdo_l", str(rdo)] if rdo else []))
    blocks = to_pixel_blocks(img)
    packed = uastc.encode_uastc_blocks(hip_ctx, blocks, level | (uastc.FAVOR_SIMPLER_MODES if rdo else 0))
    if rdo:
        packed, _ = uastc.uastc_rdo(hip_ctx, packed, blocks, uastc.RdoParams(m_lambda=rdo), level, total_jobs=1)
    mine = uastc_basis_file(packed, [(0, w // 4, h // 4, w, h, 0, 0, int(alpha))], key_values=basis_file_key_values(cli))
    assert mine.shape == cli.shape and (mine == cli).all()


@pytest.mark.skipif(not __import__("helpers").have_ref_cli(), reason="oracle/_ref/basisu not present")
@pytest.mark.parametrize("name,kw,cli_args,alpha", [
    ("etc1s_basis", dict(quality=128), ["-etc1s", "-q", "128"], False),
    ("etc1s_alpha_mips_ktx2", dict(quality=160, comp_level=2, mipmaps=True, ktx2=True), ["-etc1s", "-q", "160", "-comp_level", "2", "-mipmap"], True),
    ("uastc_rdo_mips_basis", dict(uastc=True, uastc_level=1, uastc_rdo_lambda=1.5, mipmaps=True), ["-uastc", "-uastc_level", "1", "-uastc_rdo_l", "1.5", "-mipmap"], False),
    ("uastc_alpha_ktx2", dict(uastc=True, ktx2=True), ["-uastc", "-ktx2_no_zstandard"], True),
])
def test_compress_matches_reference_command_line(hip_ctx, tmp_path, name, kw, cli_args, alpha):
    """compress(): one call from an RGBA image to the file, every stage on its MI355X path, against `basisu <options> x.png`."""
    from helpers import save_png, run_ref_cli, basis_file_key_values, ktx2_file_key_values
    from basis_universal_amd.compress import compress
    w, h = 148, 100
    img = np.ascontiguousarray(synth(148, 100, 17))
    if alpha:
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 3] = np.clip(128 + 110 * np.sin(xx / 15.0) * np.cos(yy / 12.0), 0, 255).astype(np.uint8)
    save_png(tmp_path / "x.png", img)
    ktx2 = bool(kw.get("ktx2"))
    cli = run_ref_cli(tmp_path / "x.png", *cli_args, ktx2=ktx2)
    mine = compress(hip_ctx, img, key_values=(ktx2_file_key_values if ktx2 else basis_file_key_values)(cli), **kw)
    assert mine.shape == cli.shape and (mine == cli).all(), name


@needs_ref
@pytest.mark.parametrize("level", [1, 2])
def test_video_clip_matches_reference(hip_ctx, level):
    """cBASISTexTypeVideoFrames end to end: resident frontend in video mode + backend with conditional replenishment vs the reference pair."""
    from basis_universal_amd.etc1s import Etc1sFrontend
    from basis_universal_amd.backend import Etc1sBackend
    w, h, frames = 128, 96, 3
    base = synth(w, h, 300)
    clip = []
    for f in range(frames):
        img = base.copy()
        img[24:56, 8 + 12 * f:48 + 12 * f] = synth(40, 32, 310 + f)
        clip.append(img)
    blocks = np.concatenate([to_pixel_blocks(i) for i in clip])
    nbx, nby = w // 4, h // 4
    slices = [(f * nbx * nby, nbx, nby, w, h, f, 0, 0, int(f == 0)) for f in range(frames)]
    fe = Etc1sFrontend(hip_ctx, video=True)
    fe.init(blocks, 300, 300, level, True)
    fe.compress()
    be = Etc1sBackend.from_frontend(fe, slices, 1.5, 1.25, level, video=True)
    total = be.encode()
    ref = RefFrontend(blocks, 300, 300, level, True)
    ref.set_tex_type(3)
    ref.call("compress")
    assert total == ref.backend_run([s[:3] for s in slices], 1.5, 1.25)[0]
    for k in OUTPUTS:
        a, b = be.get(k), ref.backend_get(k)
        assert a.shape == b.shape and (a == b).all(), k
    for s in range(frames):
        assert (be.get("slice_image_data", s) == ref.backend_get("slice_image_data", s)).all()
    be.close(); fe.close(); ref.close()


@pytest.mark.parametrize("w,h,seed,perceptual", [(512, 384, 21, True), (260, 132, 22, False)])
def test_device_block_errors_equal_the_host_ones(hip_ctx, w, h, seed, perceptual):
    """create_encoder_blocks' stateless part (every block's error as encoded and under its causal neighbours' endpoints) comes from k_backend_block_errors when a resident
    frontend is behind the backend, and from the host loop when the backend is driven from plain arrays: same state in, so same bytes and same per-block decisions out --
    and the kernel must really have run."""
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend
    blocks = to_pixel_blocks(synth(w, h, seed))
    nbx, nby = (w + 3) // 4, (h + 3) // 4
    ep, sel = quality_to_clusters(128, blocks.shape[0])
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, ep, sel, 1, perceptual)
    fe.compress()
    prm = fe.get("endpoint_cluster_etc_params").reshape(-1, 16)[:, :4].copy()
    arrays = dict(source_blocks=blocks, output_blocks=fe.get("encoded_blocks"), block_endpoint_index=fe.get("block_endpoint_clusters_indices", np.uint32),
                  block_selector_index=fe.get("block_selector_cluster_index", np.uint32), endpoint_color5_inten=prm, selector_blocks=fe.get("optimized_cluster_selectors"))
    host = Etc1sBackend.from_arrays(slices=[(0, nbx, nby)], perceptual=perceptual, endpoint_rdo_thresh=1.5, selector_rdo_thresh=1.25, compression_level=1, **arrays)
    hip_ctx.profile_enable(True)
    dev = Etc1sBackend.from_frontend(fe, [(0, nbx, nby)], 1.5, 1.25, 1)
    assert dev.encode() == host.encode()
    ran = hip_ctx.profile_read()
    hip_ctx.profile_enable(False)
    assert "backend_block_errors" in ran, sorted(ran)
    for k in OUTPUTS + ["encoder_blocks", "endpoint_remap_old_to_new", "selector_remap_new_to_old"]:
        a, b = dev.get(k), host.get(k)
        assert a.shape == b.shape and (a == b).all(), k
    a, b = dev.get("slice_image_data", 0), host.get("slice_image_data", 0)
    assert a.shape == b.shape and (a == b).all()
    dev.close(); host.close(); fe.close()
