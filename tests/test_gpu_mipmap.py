"""-m gpu: mip generation on the device against the real image_resample / the reference tool: the HIP kernels apply the
host-built resampling plan (tests/test_mipmap_host.py pins the plan itself on the CPU) and have to write the reference's bytes."""
import numpy as np
import pytest

from helpers import have_ref, have_ref_cli, synth, to_pixel_blocks

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")


@needs_ref
def test_resample_matches_image_resample(hip_ctx):
    from basis_universal_amd import mipmap
    from test_mipmap_host import CASES, rgba, reference
    for sw, sh, dw, dh, srgb, flt, scale, wrap, comps in CASES:
        for seed, noise in ((1, False), (2, True)):
            src = rgba(sw, sh, seed, noise)
            got = mipmap.resample(hip_ctx, src, dw, dh, srgb, flt, scale, wrap, comps)
            exp = reference(src, dw, dh, srgb, flt, scale, wrap, comps)
            assert (got == exp).all(), ((sw, sh, dw, dh, srgb, flt, scale, wrap, comps), np.argwhere(got != exp)[:5])


@needs_ref
@pytest.mark.parametrize("w,h,alpha", [(512, 384, False), (301, 173, True), (1024, 1024, True)])
def test_mip_chain_matches_reference(hip_ctx, w, h, alpha):
    from basis_universal_amd import mipmap
    from test_mipmap_host import rgba
    from test_backend_host import _ref_mip_chain
    img = rgba(w, h, 7)
    if not alpha:
        img[..., 3] = 255
    got = mipmap.generate_mipmaps(hip_ctx, img, has_alpha=alpha)
    exp = _ref_mip_chain(img, alpha)
    assert [g.shape for g in got] == [e.shape for e in exp] and got[-1].shape[:2] == (1, 1)
    for level, (g, e) in enumerate(zip(got, exp)):
        assert (g == e).all(), (level, np.argwhere(g != e)[:5])


@pytest.mark.skipif(not have_ref_cli(), reason="oracle/_ref/basisu not present")
@pytest.mark.parametrize("w,h,alpha", [(256, 192, False), (100, 60, True)])
def test_mipmapped_encode_matches_reference_command_line(hip_ctx, tmp_path, w, h, alpha):
    """`basisu -etc1s -q 128 -mipmap x.png`, .basis and .ktx2, against: mips on the device -> tiles -> resident frontend -> backend -> writers."""
    from helpers import save_png, run_ref_cli, basis_file_key_values, ktx2_file_key_values
    from basis_universal_amd import mipmap
    from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
    from basis_universal_amd.backend import Etc1sBackend, default_params
    img = np.ascontiguousarray(synth((w + 3) // 4 * 4, (h + 3) // 4 * 4, 91)[:h, :w])
    if alpha:
        yy, xx = np.mgrid[0:h, 0:w]
        img[..., 3] = np.clip(128 + 100 * np.sin(xx / 13.0) * np.cos(yy / 11.0), 0, 255).astype(np.uint8)
    save_png(tmp_path / "x.png", img)
    levels = [img] + mipmap.generate_mipmaps(hip_ctx, img, has_alpha=alpha)
    blocks, slices, first = [], [], 0
    for mip, lv in enumerate(levels):
        lh, lw = lv.shape[:2]
        nbx, nby = (lw + 3) // 4, (lh + 3) // 4
        planes = [lv]
        if alpha:
            rgb = lv.copy(); rgb[..., 3] = 255
            a = np.repeat(lv[..., 3:4], 4, axis=2); a[..., 3] = 255
            planes = [rgb, a]
        for k, pl in enumerate(planes):
            blocks.append(to_pixel_blocks(pl))
            slices.append((first, nbx, nby, lw, lh, 0, mip, k))
            first += nbx * nby
    blocks = np.concatenate(blocks)
    max_ep, max_sel = quality_to_clusters(128, blocks.shape[0])
    fe = Etc1sFrontend(hip_ctx)
    fe.init(blocks, max_ep, max_sel, 1, True)
    fe.compress()
    ept, selt = default_params(128, 1)
    be = Etc1sBackend.from_frontend(fe, slices, ept, selt, 1)
    be.encode()
    cli = run_ref_cli(tmp_path / "x.png", "-etc1s", "-q", "128", "-mipmap")
    mine = be.basis_file(key_values=basis_file_key_values(cli))
    assert mine.shape == cli.shape and (mine == cli).all()
    cli2 = run_ref_cli(tmp_path / "x.png", "-etc1s", "-q", "128", "-mipmap", ktx2=True)
    mine2 = be.ktx2_file(has_alpha=alpha, key_values=ktx2_file_key_values(cli2))
    assert mine2.shape == cli2.shape and (mine2 == cli2).all()
    be.close(); fe.close()
