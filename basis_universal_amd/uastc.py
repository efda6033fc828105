"""UASTC LDR 4x4 block encoding on the GPU: the host-side mirror of basisu::encode_uastc (encoder/basisu_uastc_enc.h:69) as a batch op.

`encode_uastc_blocks` takes what the reference's call site has at hand (comp.cpp:2020-2033): an array of 4x4 RGBA source blocks and the
pack flags (cPackUASTCLevel* | cPackUASTCFavor* | cPackUASTCETC1*). It returns one 16-byte basist::uastc_block per source block,
bit-identical to the reference's. There is no CPU implementation here: without the HIP library and a GPU it raises.
"""
import ctypes as C

import numpy as np

from . import capi

# pack flags, same values as encoder/basisu_uastc_enc.h:24-66
LEVEL_FASTEST, LEVEL_FASTER, LEVEL_DEFAULT, LEVEL_SLOWER, LEVEL_VERY_SLOW = range(5)
FAVOR_UASTC_ERROR = 8
FAVOR_BC7_ERROR = 16
ETC1_FASTER_HINTS = 64
ETC1_FASTEST_HINTS = 128
ETC1_DISABLE_FLIP_AND_INDIVIDUAL = 256
FAVOR_SIMPLER_MODES = 512


def encode_uastc_blocks(ctx, blocks, flags=LEVEL_DEFAULT, n_blocks=None, out_device=None):
    """blocks: (n, 4, 4, 4) / (n, 64) uint8 array (uploaded once) or a device pointer (int) to n_blocks resident tiles.
    Returns an (n, 16) uint8 array, or None when out_device (device pointer for n*16 bytes) is given."""
    own = None
    if isinstance(blocks, np.ndarray):
        blocks = np.ascontiguousarray(blocks, np.uint8)
        n = blocks.size // 64
        d_px = own = ctx.upload(blocks)
    else:
        if n_blocks is None:
            raise ValueError("n_blocks is required with a device pointer")
        n, d_px = int(n_blocks), blocks
    d_out = out_device if out_device is not None else ctx.alloc(max(n * 16, 1))
    try:
        if n:
            ctx.check(ctx.lib.k_encode_uastc_blocks(ctx.h, C.c_void_p(d_px), n, int(flags), C.c_void_p(d_out)), "encode_uastc_blocks")
        if out_device is not None:
            return None
        return ctx.download(d_out, (n, 16), np.uint8)
    finally:
        if own is not None:
            ctx.free(own)
        if out_device is None:
            ctx.free(d_out)
