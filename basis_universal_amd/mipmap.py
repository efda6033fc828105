"""Mip generation on the device: basis_compressor::generate_mipmaps (encoder/basisu_comp.cpp:2146-2230) as a chain of
bu_generate_mipmap_level calls -- contributor lists on the host exactly as the reference's Resampler builds them, pixels resampled by the HIP
kernels of mipmap_kernels.hip. Byte-identical to image_resample."""
import ctypes as C

import numpy as np

from . import capi
from .etc1s import load_frontend_library

_vp = C.c_void_p


def _lib():
    L = load_frontend_library()
    if not getattr(L, "_mip_bound", False):
        L.bu_generate_mipmap_level.restype = C.c_int
        L.bu_generate_mipmap_level.argtypes = [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_float, C.c_int, C.c_uint32]
        L.bu_mipmap_level_sizes.restype = C.c_uint32
        L.bu_mipmap_level_sizes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32]
        L._mip_bound = True
    return L


def level_sizes(w, h, smallest_dimension=1):
    out = np.zeros(64, np.uint32)
    n = _lib().bu_mipmap_level_sizes(w, h, smallest_dimension, out.ctypes.data_as(_vp), 32)
    return [tuple(int(v) for v in out[2 * i:2 * i + 2]) for i in range(n)]


def resample(ctx, image, dst_w, dst_h, srgb=True, filter="kaiser", filter_scale=1.0, wrapping=True, num_comps=4):
    """image_resample on the GPU: (h, w, 4) u8 -> (dst_h, dst_w, 4) u8."""
    img = np.ascontiguousarray(image, np.uint8)
    h, w = img.shape[:2]
    d_src, d_dst = ctx.upload(img), ctx.alloc(dst_w * dst_h * 4)
    try:
        ctx.check(_lib().bu_generate_mipmap_level(ctx.h, d_src, w, h, d_dst, dst_w, dst_h, int(srgb), filter.encode(), filter_scale, int(wrapping), num_comps),
                  "bu_generate_mipmap_level")
        return ctx.download(d_dst, (dst_h, dst_w, 4), np.uint8)
    finally:
        ctx.free(d_src); ctx.free(d_dst)


def generate_mipmaps(ctx, image, has_alpha=False, srgb=True, filter="kaiser", filter_scale=1.0, wrapping=True, smallest_dimension=1, fast=True):
    """The levels below `image` ((h, w, 4) u8) with the compressor's defaults; with `fast` (m_mip_fast) every level past the first is made from
    the one above it, all on the device: one upload, one download per level."""
    img = np.ascontiguousarray(image, np.uint8)
    h, w = img.shape[:2]
    sizes = level_sizes(w, h, smallest_dimension)
    bufs = [(ctx.upload(img), w, h)]
    out = []
    try:
        for lw, lh in sizes:
            src, sw, sh = bufs[-1] if fast else bufs[0]
            d = ctx.alloc(lw * lh * 4)
            bufs.append((d, lw, lh))
            ctx.check(_lib().bu_generate_mipmap_level(ctx.h, src, sw, sh, d, lw, lh, int(srgb), filter.encode(), filter_scale, int(wrapping), 4 if has_alpha else 3),
                      "bu_generate_mipmap_level")
            out.append(ctx.download(d, (lh, lw, 4), np.uint8))
    finally:
        for d, _, _ in bufs:
            ctx.free(d)
    return out
