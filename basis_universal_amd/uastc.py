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


class RdoParams(C.Structure):
    """uastc_rdo_params (encoder/basisu_uastc_enc.h:94-134), same fields, same defaults (= bu_uastc_rdo_params in include/basisu_hip.h)."""
    _fields_ = [("m_lambda", C.c_float), ("m_max_allowed_rms_increase_ratio", C.c_float), ("m_skip_block_rms_thresh", C.c_float),
                ("m_max_smooth_block_std_dev", C.c_float), ("m_smooth_block_max_error_scale", C.c_float),
                ("m_lz_dict_size", C.c_uint32), ("m_lz_literal_cost", C.c_uint32), ("m_endpoint_refinement", C.c_uint32)]

    def __init__(self, **kw):
        super().__init__()
        self.m_lz_dict_size, self.m_lambda, self.m_max_allowed_rms_increase_ratio, self.m_skip_block_rms_thresh = 4096, 0.5, 10.0, 8.0
        self.m_endpoint_refinement, self.m_lz_literal_cost = 1, 100
        self.m_max_smooth_block_std_dev, self.m_smooth_block_max_error_scale = 18.0, 10.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


def uastc_rdo(ctx, uastc_blocks, pixel_blocks, params=None, flags=LEVEL_DEFAULT, total_jobs=0, n_blocks=None):
    """basisu::uastc_rdo (uastc_enc.h:139) on the GPU. uastc_blocks / pixel_blocks: numpy arrays ((n,16) and (n,4,4,4) uint8; a modified copy
    of the blocks is returned with the stats) or device pointers (ints; the blocks are modified in place, n_blocks required).
    total_jobs as in the reference: 0/1 = one strip, k = strips of n // k blocks (comp.cpp:2078 uses min(4, threads)).
    Returns (blocks or None, {"modified", "refined", "skipped", "strips"})."""
    params = params or RdoParams()
    host = isinstance(uastc_blocks, np.ndarray)
    owned = []
    try:
        if host:
            src = np.ascontiguousarray(uastc_blocks, np.uint8)
            n = src.size // 16
            d_blk = ctx.upload(src) if n else 0
            if n:
                owned.append(d_blk)
        else:
            if n_blocks is None:
                raise ValueError("n_blocks is required with device pointers")
            n, d_blk = int(n_blocks), uastc_blocks
        if isinstance(pixel_blocks, np.ndarray):
            px = np.ascontiguousarray(pixel_blocks, np.uint8)
            if px.size != n * 64:
                raise ValueError("pixel_blocks must hold 64 bytes per UASTC block")
            d_px = ctx.upload(px) if n else 0
            if n:
                owned.append(d_px)
        else:
            d_px = pixel_blocks
        stats = (C.c_uint32 * 4)()
        if n:
            ctx.check(ctx.lib.k_uastc_rdo(ctx.h, C.c_void_p(d_blk), C.c_void_p(d_px), n, C.byref(params), int(flags), int(total_jobs), stats), "uastc_rdo")
        info = {"modified": stats[0], "refined": stats[1], "skipped": stats[2], "strips": stats[3]}
        if host:
            return (ctx.download(d_blk, (n, 16), np.uint8) if n else np.zeros((0, 16), np.uint8)), info
        return None, info
    finally:
        for d in owned:
            ctx.free(d)


class UastcPipeline:
    """encode_uastc (+ uastc_rdo) over a stream of images with several in flight on one GPU (bu_hip_uastc_pipeline_*, include/basisu_hip.h): every submission is
    enqueued on one of `lanes` private streams without a host synchronisation, so one image's RDO walk (a serial chain per strip that leaves most of the chip idle)
    runs beside the next image's encode kernels. Device pointers in, device pointers out; bytes as encode_uastc_blocks followed by uastc_rdo."""

    def __init__(self, ctx, lanes, max_blocks, flags=LEVEL_DEFAULT, max_total_jobs=4):
        self.ctx = ctx
        self.h = ctx.lib.uastc_pipeline_create(ctx.h, int(lanes), int(max_blocks), int(flags), int(max_total_jobs))
        if not self.h:
            raise capi.HipError(f"uastc_pipeline_create failed: {ctx.lib.last_error(ctx.h)}")

    def submit(self, d_px, n_blocks, d_out, rdo_params=None, flags=LEVEL_DEFAULT, total_jobs=0):
        """-> ticket. rdo_params: an RdoParams to run the post-pass, None for the plain encode."""
        ticket = C.c_uint64()
        self._keep = rdo_params   # the structure is read during the call only; kept for symmetry with the other wrappers
        self.ctx.check(self.ctx.lib.uastc_pipeline_submit(self.h, C.c_void_p(d_px), int(n_blocks), C.c_void_p(d_out), C.byref(rdo_params) if rdo_params is not None else None,
                                                          int(flags), int(total_jobs), C.byref(ticket)), "uastc_pipeline_submit")
        return ticket.value

    def wait(self, ticket=0):
        """ticket 0: everything submitted so far. Returns the ticket's RDO statistics (zeros for ticket 0)."""
        stats = (C.c_uint32 * 4)()
        self.ctx.check(self.ctx.lib.uastc_pipeline_wait(self.h, int(ticket), stats), "uastc_pipeline_wait")
        return {"modified": stats[0], "refined": stats[1], "skipped": stats[2], "strips": stats[3]}

    def close(self):
        if self.h:
            self.ctx.lib.uastc_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
