"""One call from an image to a file: the C-style entry of the reference (`basis_compress`, encoder/basisu_comp.cpp:5561-5900 ->
basis_compressor::process, comp.cpp:619-1040) for the two hot paths of this package, every stage in its MI355X-native form:

    raster in HBM -> mip levels (mipmap_kernels.hip) -> 4x4 tiles (k_extract_blocks)
      ETC1S : resident frontend (etc1s.Etc1sFrontend) -> host backend (backend.Etc1sBackend) -> .basis / .ktx2
      UASTC : encode_uastc kernels (-> uastc_rdo kernels) -> .basis / .ktx2 (no Zstandard)

The result is the file the reference command line tool writes for the same options (tests/test_gpu_backend.py, test_gpu_mipmap.py), given the
same key-values. No stage has a CPU implementation here: without the HIP libraries and a GPU the context cannot be created."""
import numpy as np

from . import mipmap, uastc as _uastc
from .backend import Etc1sBackend, default_params, uastc_basis_file, uastc_ktx2_file
from .etc1s import Etc1sFrontend, quality_to_clusters


def compress(ctx, image, *, uastc=False, quality=128, comp_level=1, uastc_level=_uastc.LEVEL_DEFAULT, uastc_rdo_lambda=None, uastc_rdo_jobs=1, mipmaps=False,
             ktx2=False, srgb=True, key_values=()):
    """image: (h, w, 4) uint8 RGBA. Returns the file as a uint8 array.
    ETC1S: quality 1-255 (`-q`), comp_level 0-6 (`-comp_level`). UASTC: uastc_level 0-4, uastc_rdo_lambda (`-uastc_rdo_l`) or None.
    mipmaps: the compressor's defaults (Kaiser, sRGB-aware, wrapping, down to 1x1). srgb: perceptual metrics + sRGB transfer function flag."""
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("image must be (h, w, 4) uint8")
    h, w = img.shape[:2]
    has_alpha = bool((img[..., 3] != 255).any())                      # image::has_alpha -> m_any_source_image_has_alpha
    # ---- the levels, resident
    sizes = [(w, h)] + (mipmap.level_sizes(w, h) if mipmaps else [])
    rasters = [ctx.upload(img)]
    owned = list(rasters)
    try:
        for lw, lh in sizes[1:]:
            d = ctx.alloc(lw * lh * 4)
            owned.append(d)
            sw, sh = sizes[len(rasters) - 1]
            ctx.check(mipmap._lib().bu_generate_mipmap_level(ctx.h, rasters[-1], sw, sh, d, lw, lh, int(srgb), b"kaiser", 1.0, 1, 4 if has_alpha else 3),
                      "bu_generate_mipmap_level")
            rasters.append(d)
        # ---- the slices: one per level, for ETC1S with alpha a colour slice and an (a, a, a) slice per level (comp.cpp:2880-2910)
        split_alpha = has_alpha and not uastc
        per_level = [((lw + 3) // 4) * ((lh + 3) // 4) for lw, lh in sizes]
        total_blocks = sum(per_level) * (2 if split_alpha else 1)
        d_all = ctx.alloc(total_blocks * 64)   # one contiguous tile array: the levels (and the alpha slices) share the codebooks
        owned.append(d_all)
        slices, slice_blocks, first = [], [], 0
        for mip, ((lw, lh), d_raster) in enumerate(zip(sizes, rasters)):
            nbx, nby = (lw + 3) // 4, (lh + 3) // 4
            if split_alpha:
                lv = ctx.download(d_raster, (lh, lw, 4), np.uint8)
                a = np.repeat(lv[..., 3:4], 4, axis=2); a[..., 3] = 255
                lv[..., 3] = 255
                planes = [ctx.upload(np.ascontiguousarray(lv)), ctx.upload(np.ascontiguousarray(a))]
                owned.extend(planes)
            else:
                planes = [d_raster]
            for k, d_plane in enumerate(planes):   # basis_compressor::extract_source_blocks on the resident plane, straight into its place
                ctx.check(ctx.lib.k_extract_blocks(ctx.h, d_plane, lw, lh, lw * 4, d_all + first * 64), "k_extract_blocks")
                slices.append((first, nbx, nby, lw, lh, 0, mip, k if split_alpha else int(has_alpha)))
                slice_blocks.append(nbx * nby)
                first += nbx * nby
        # ---- encode
        if uastc:
            flags = int(uastc_level) | (_uastc.FAVOR_SIMPLER_MODES if uastc_rdo_lambda else 0)       # comp.cpp:2016-2018
            d_out = ctx.alloc(total_blocks * 16)
            owned.append(d_out)
            _uastc.encode_uastc_blocks(ctx, d_all, flags, n_blocks=total_blocks, out_device=d_out)
            if uastc_rdo_lambda:
                at = 0
                for n in slice_blocks:   # the post-pass runs per slice (comp.cpp:2066-2082)
                    _uastc.uastc_rdo(ctx, d_out + at * 16, d_all + at * 64, _uastc.RdoParams(m_lambda=float(uastc_rdo_lambda)), int(uastc_level), uastc_rdo_jobs, n_blocks=n)
                    at += n
            packed = ctx.download(d_out, (total_blocks, 16), np.uint8)
            if ktx2:
                return uastc_ktx2_file(packed, slices, srgb=srgb, has_alpha=has_alpha, key_values=key_values)
            # encode_slices_to_uastc_4x4_ldr (comp.cpp:1973-1985) never sets basisu_backend_output::m_srgb, which basisu_backend_output::clear() leaves true
            # (backend.h:243): the reference's UASTC .basis files carry the sRGB header flag whatever -linear says (the .ktx2 DFD does follow the option)
            return uastc_basis_file(packed, slices, srgb=True, key_values=key_values)
        max_ep, max_sel = quality_to_clusters(quality, total_blocks)
        fe = Etc1sFrontend(ctx)
        try:
            fe.init(d_all, max_ep, max_sel, comp_level, srgb, n_blocks=total_blocks)
            fe.compress()
            ept, selt = default_params(quality, comp_level)
            be = Etc1sBackend.from_frontend(fe, slices, ept, selt, comp_level)
            try:
                be.encode()
                return be.ktx2_file(has_alpha=has_alpha, key_values=key_values) if ktx2 else be.basis_file(key_values=key_values)
            finally:
                be.close()
        finally:
            fe.close()
    finally:
        for d in owned:
            ctx.free(d)
