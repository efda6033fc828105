"""One call from an image to a file: the C-style entry of the reference (`basis_compress`, encoder/basisu_comp.cpp:5561-5900 ->
basis_compressor::process, comp.cpp:619-1040) for the two hot paths of this package, every stage in its MI355X-native form:

    raster in HBM -> mip levels (mipmap_kernels.hip) -> 4x4 tiles (k_extract_blocks)
      ETC1S : resident frontend (etc1s.Etc1sFrontend) -> host backend (backend.Etc1sBackend) -> .basis / .ktx2
      UASTC : encode_uastc kernels (-> uastc_rdo kernels) -> .basis / .ktx2 (KTX2_SS_NONE: what the reference's library default,
              comp.h:323, and `basisu -ktx2_no_zstandard` write; Zstandard supercompression of the levels is not part of this package)

The result is the file the reference command line tool writes for the same options (tests/test_gpu_backend.py, test_gpu_mipmap.py), given the
same key-values. No stage has a CPU implementation here: without the HIP libraries and a GPU the context cannot be created."""
import numpy as np

from . import mipmap, uastc as _uastc
from .backend import Etc1sBackend, default_params, uastc_basis_file, uastc_ktx2_file
from .etc1s import Etc1sFrontend, quality_to_clusters


def unified_quality_effort(uastc, quality=-1, effort=-1):
    """basis_compressor_params::set_format_mode_and_quality_effort (comp.cpp:76-92, 158-205; `basisu -quality Q -effort E`) for the two LDR formats here.
    quality in [1, 100] (-1: leave the default), effort in [0, 10] (-1: default). Returns the low-level settings as keyword arguments of compress():
      ETC1S : quality = round(255 q / 100), comp_level = round(6 e / 10)                       (std::round: halves away from zero)
      UASTC : uastc_level = round(4 e / 10); quality < 100 switches the RDO post-pass on with lambda = 20 (1 - q/100)^1.3 in binary32
              (uastc_ldr_4x4_lambda_from_quality, comp.cpp:54-63), quality 100 switches it off."""
    def rnd(x):
        return int(np.floor(np.float32(x) + np.float32(0.5)))
    if quality > 0:
        quality = min(max(int(quality), 0), 100)
    if effort > 0:
        effort = min(max(int(effort), 0), 10)
    fq = np.float32(min(max(np.float32(quality) / np.float32(100.0), np.float32(0)), np.float32(1))) if quality >= 0 else np.float32(0)
    fe = np.float32(min(max(np.float32(effort) / np.float32(10.0), np.float32(0)), np.float32(1))) if effort >= 0 else np.float32(0)
    lerp = lambda a, b, c: np.float32(a) + (np.float32(b) - np.float32(a)) * np.float32(c)       # basisu::lerp, enc.h
    if not uastc:
        out = {}
        if quality >= 0:
            out["quality"] = rnd(lerp(0, 255.0, fq))
        out["comp_level"] = rnd(lerp(0, 6.0, fe)) if effort >= 0 else 2      # BASISU_DEFAULT_ETC1S_COMPRESSION_LEVEL (the library default; the CLI's is 1)
        return out
    out = {"uastc": True, "uastc_level": rnd(lerp(0, 4.0, fe)) if effort >= 0 else _uastc.LEVEL_DEFAULT, "uastc_rdo_lambda": None}
    if 0 <= quality < 100:
        # `pow(1.0f - q, 1.3f)` is the binary32 overload; the correctly rounded double power rounded once more agrees with glibc's powf for every
        # quality 0..99 (tests/test_host_logic.py holds all hundred to the reference's own values)
        out["uastc_rdo_lambda"] = float(np.float32(20.0) * np.float32(float(np.float32(1.0) - fq) ** float(np.float32(1.3)))) if fq < 1 else 0.0
    return out


def compress(ctx, image, *, uastc=False, quality=128, comp_level=1, uastc_level=_uastc.LEVEL_DEFAULT, uastc_rdo_lambda=None, uastc_rdo_jobs=1, mipmaps=False,
             ktx2=False, srgb=True, key_values=(), max_threads=0):
    """image: (h, w, 4) uint8 RGBA. Returns the file as a uint8 array.
    ETC1S: quality 1-255 (`-q`), comp_level 0-6 (`-comp_level`). UASTC: uastc_level 0-4, uastc_rdo_lambda (`-uastc_rdo_l`; None = no post-pass, any float
    incl. 0.0 = post-pass on, as m_rdo_uastc_ldr_4x4 + its scalar), uastc_rdo_jobs = the strips of the post-pass (the reference: min(4, pool threads) when
    multithreaded, else 1; comp.cpp:2078). `**unified_quality_effort(...)` gives the settings of `-quality` / `-effort`.
    max_threads: the reference's codebook-thread configuration (0 / 1 = `-no_multithreading`; T > 1 = the T-way partitioned codebook build its
    multi-threaded default takes from 262,144 distinct training vectors up, enc.h:2086-2215: etc1s.reference_max_threads() gives the T a host would use).
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
            rdo = uastc_rdo_lambda is not None and uastc_rdo_lambda is not False
            flags = int(uastc_level) | (_uastc.FAVOR_SIMPLER_MODES if rdo else 0)       # comp.cpp:2016-2018
            d_out = ctx.alloc(total_blocks * 16)
            owned.append(d_out)
            _uastc.encode_uastc_blocks(ctx, d_all, flags, n_blocks=total_blocks, out_device=d_out)
            if rdo:
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
        fe = Etc1sFrontend(ctx, max_threads=max_threads)
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
