"""Shared test plumbing: ctypes bindings for the CPU checkers and the synthetic inputs.

Nothing here touches the product; see basis_universal_amd/ for that. The two checkers are
  * oracle/liboracle_etc1s.so   -- our plain-C restatement (travels to the GPU box as source + .so)
  * oracle/_ref/libref_harness.so -- the REAL reference compiled from /root/reference (prebuilt; travels as a binary)
"""
import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
REF_DIR = pathlib.Path("/root/reference")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


def ptr(a, t=u8p):
    return a.ctypes.data_as(t) if a is not None else None


# ----------------------------------------------------------------------------- inputs

def synth(w, h, seed):
    """SURVEY.md §8(d) synthetic RGBA image: smooth sinusoids + 4x4 tile noise N(0,12) + pixel noise N(0,6)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 3), np.float32)
    img[..., 0] = 128 + 100 * np.sin(x / 97) * np.cos(y / 131)
    img[..., 1] = 128 + 90 * np.sin((x + y) / 211)
    img[..., 2] = 128 + 110 * np.cos(x / 53 + y / 71)
    tile = rng.normal(0, 12, (h // 4, w // 4, 3)).astype(np.float32)
    img += np.repeat(np.repeat(tile, 4, axis=0), 4, axis=1)
    img += rng.normal(0, 6, (h, w, 3)).astype(np.float32)
    out = np.empty((h, w, 4), np.uint8)
    out[..., :3] = np.clip(img, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def uniform_random(w, h, seed=42):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    out[..., 3] = 255
    return out


def endpoint_cube(w, h, seed=7):
    """The worst case for the ENDPOINT codebook builder: block b is drawn from the four colours of ETC1S endpoint (colour5, intensity table) number perm[b mod 2^18] with
    random selectors, so the per-block fit finds (nearly) every one of the 32^3 x 8 endpoints and the builder gets as many distinct training vectors as an image can give it
    (236,235 is the ceiling, tests/test_host_logic.py::test_endpoint_codebook_can_never_reach_the_threaded_gate; 14,177 for uniform noise, 35,502 for the synthetic image)."""
    rng = np.random.default_rng(seed)
    nb = (h // 4) * (w // 4)
    code = rng.permutation(1 << 18)[np.arange(nb) % (1 << 18)]
    table = np.array([[-8, -2, 2, 8], [-17, -5, 5, 17], [-29, -9, 9, 29], [-42, -13, 13, 42], [-60, -18, 18, 60], [-80, -24, 24, 80], [-106, -33, 33, 106], [-183, -47, 47, 183]], np.int32)
    c5 = np.stack([(code >> 10) & 31, (code >> 5) & 31, code & 31], -1).astype(np.int32)
    c8 = (c5 << 3) | (c5 >> 2)
    sel = rng.integers(0, 4, (nb, 16))
    d = table[(code >> 15) & 7][np.arange(nb)[:, None], sel]                       # (nb, 16)
    px = np.clip(c8[:, None, :] + d[:, :, None], 0, 255).astype(np.uint8)          # (nb, 16, 3)
    out = np.empty((h, w, 4), np.uint8)
    out[..., 3] = 255
    out[..., :3] = px.reshape(h // 4, w // 4, 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(h, w, 3)
    return out


def kodak_mosaic(w=4096, h=4096):
    """The 24 Kodak images of the reference's own test set (tests/golden/kodak24.npz: the pixels of test_files/kodim01-24.png) tiled into one w x h RGBA image:
    photographic statistics at BASELINE.json's full size. 768 x 512 slots, portrait images transposed; slot i holds image i mod 24, mirrored top to bottom from
    the second time round (so no block repeats); cropped to w x h."""
    z = np.load(pathlib.Path(__file__).parent / "golden" / "kodak24.npz")
    imgs = [z[f"k{i:02d}"] for i in range(1, 25)]
    imgs = [im if im.shape[0] == 512 else np.ascontiguousarray(im.transpose(1, 0, 2)) for im in imgs]
    cols, rows = (w + 767) // 768, (h + 511) // 512
    out = np.empty((rows * 512, cols * 768, 4), np.uint8)
    out[..., 3] = 255
    for i in range(rows * cols):
        im = imgs[i % 24]
        if (i // 24) & 1:
            im = im[::-1]
        r, c = divmod(i, cols)
        out[r * 512:(r + 1) * 512, c * 768:(c + 1) * 768, :3] = im
    return np.ascontiguousarray(out[:h, :w])


def to_pixel_blocks(img):
    """(H, W, 4) u8 -> (n_blocks, 4, 4, 4) u8 in block-raster order, [y][x] inside the block (comp.cpp:3207-3268).
    Edges are clamped like image::extract_block_clamped."""
    h, w, _ = img.shape
    bh, bw = (h + 3) // 4, (w + 3) // 4
    if (h % 4) or (w % 4):
        ys = np.minimum(np.arange(bh * 4), h - 1)
        xs = np.minimum(np.arange(bw * 4), w - 1)
        img = img[ys][:, xs]
    blocks = img.reshape(bh, 4, bw, 4, 4).transpose(0, 2, 1, 3, 4)
    return np.ascontiguousarray(blocks.reshape(bh * bw, 4, 4, 4))


def load_png(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGBA"), dtype=np.uint8))


def csr_from_lists(lists):
    offs = np.zeros(len(lists) + 1, np.uint32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    idx = np.concatenate([np.asarray(l, np.uint32) for l in lists]) if len(lists) and offs[-1] else np.zeros(0, np.uint32)
    return offs, np.ascontiguousarray(idx, np.uint32)


def csr_blob_split(blob):
    """[n, off_0..off_n, idx...] u32 blob (oracle/ref_harness.cpp csr()) -> (offsets, indices)."""
    n = int(blob[0])
    offs = blob[1:n + 2].copy()
    idx = blob[n + 2:n + 2 + int(offs[-1])].copy()
    return offs, idx


# ----------------------------------------------------------------------------- C oracle

_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        so = ORACLE_DIR / "liboracle_etc1s.so"
        src = ORACLE_DIR / "etc1s_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.check_call(["make", "-C", str(ORACLE_DIR), "liboracle_etc1s.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(str(so))
        L.orc_color_distance.restype = C.c_uint32
        L.orc_color_distance.argtypes = [C.c_int, u8p, u8p]
        L.orc_hash_hsieh3.restype = C.c_uint32
        L.orc_hash_hsieh3.argtypes = [C.c_uint8] * 3
        L.orc_etc1_optimize.restype = C.c_int
        L.orc_etc1_optimize.argtypes = [u8p, C.c_uint32, C.c_int, C.c_int, u8p, u32p, u64p, u8p]
        L.orc_encode_etc1s_blocks.argtypes = [u8p, C.c_uint32, C.c_int, C.c_int, u8p]
        L.orc_determine_selectors.argtypes = [u8p, C.c_uint32, u8p, C.c_int, u8p]
        L.orc_generate_endpoint_codebook.argtypes = [u8p, C.c_uint32, u32p, u32p, C.c_int, C.c_int, C.c_uint32, u8p, u64p, u8p]
        L.orc_refine_endpoint_clusterization.argtypes = [u8p, C.c_uint32, u32p, u8p, C.c_uint32, C.c_uint32, u32p, u32p, u8p, C.c_int, u32p]
        L.orc_create_optimized_selector_codebook.argtypes = [u8p, u8p, C.c_uint32, u32p, u32p, C.c_int, u8p]
        L.orc_find_optimal_selector_clusters.argtypes = [u8p, u8p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, u32p, u32p, u8p, C.c_int, C.c_uint32, u32p]
        L.orc_endpoint_training_vectors.argtypes = [u8p, C.c_uint32, f32p]
        L.orc_selector_training_vectors.argtypes = [u8p, C.c_uint32, C.c_int, f32p, u64p]
        _oracle = L
    return _oracle


def orc_encode_blocks(blocks, level, perceptual=True):
    n = blocks.shape[0]
    out = np.zeros((n, 8), np.uint8)
    oracle().orc_encode_etc1s_blocks(ptr(blocks), n, level, int(perceptual), ptr(out))
    return out


# ----------------------------------------------------------------------------- the real reference

_ref = None


def have_ref():
    return (ORACLE_DIR / "_ref" / "libref_harness.so").exists()


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(str(ORACLE_DIR / "_ref" / "libref_harness.so"))
        L.ref_init.restype = C.c_int
        L.ref_etc1_optimize.restype = C.c_int
        L.ref_etc1_optimize.argtypes = [u8p, C.c_uint32, C.c_int, C.c_int, u8p, u32p, u64p, u8p]
        L.ref_encode_etc1s_blocks.argtypes = [u8p, C.c_uint32, C.c_int, C.c_int, u8p]
        L.ref_determine_selectors.argtypes = [u8p, C.c_uint32, u8p, C.c_int, u8p]
        L.ref_color_distance.restype = C.c_uint32
        L.ref_color_distance.argtypes = [C.c_int, u8p, u8p]
        L.ref_frontend_create.restype = C.c_void_p
        L.ref_frontend_create.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        L.ref_frontend_destroy.argtypes = [C.c_void_p]
        L.ref_frontend_call.restype = C.c_int64
        L.ref_frontend_call.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.ref_frontend_get.restype = C.c_uint64
        L.ref_frontend_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.ref_tsvq.restype = C.c_int
        L.ref_tsvq.argtypes = [C.c_uint32, f32p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u32p, C.c_uint64, u32p, C.c_uint64]
        L.ref_tsvq_mt.restype = C.c_int
        L.ref_tsvq_mt.argtypes = [C.c_uint32, f32p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u32p, C.c_uint64, u32p, C.c_uint64]
        L.ref_encode_uastc.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.ref_uastc_rdo.restype = C.c_int
        L.ref_uastc_rdo.argtypes = [u8p, u8p, C.c_uint32, f32p, u32p, C.c_uint32, C.c_uint32]
        L.ref_color_cell_compression.restype = C.c_uint64
        L.ref_color_cell_compression.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, u8p, u8p]
        L.ref_ccell_est.restype = C.c_uint64
        L.ref_ccell_est.argtypes = [C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_uint64]
        L.ref_table.restype = C.c_uint64
        L.ref_table.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        L.ref_compress_etc1s.restype = C.c_int
        L.ref_compress_etc1s.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, u32p, u32p, u8p, C.c_uint64, u64p]
        L.ref_quality_effort.restype = C.c_int
        L.ref_quality_effort.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), f32p]
        assert L.ref_init() == 1
        _ref = L
    return _ref


def ref_quality_effort(uastc, quality, effort):
    """basis_compressor_params::set_format_mode_and_quality_effort of the reference -> (etc1s quality, etc1s comp level, uastc pack flags, rdo flag, lambda as f32)"""
    out, lam = np.zeros(4, np.int32), np.zeros(1, np.float32)
    assert ref().ref_quality_effort(int(uastc), quality, effort, out.ctypes.data_as(C.POINTER(C.c_int32)), ptr(lam, f32p)) == 1
    return int(out[0]), int(out[1]), int(out[2]), bool(out[3]), lam[0]


class RefFrontend:
    """The reference's basisu_frontend, driven one private stage at a time (oracle/ref_harness.cpp)."""

    def __init__(self, blocks, max_ep, max_sel, level, perceptual=True, threads=1):
        """threads: the reference's job pool size including the caller; 1 is the pinned configuration every parity check uses"""
        self.L = ref()
        self.blocks = np.ascontiguousarray(blocks)
        if threads == 1:
            self.h = self.L.ref_frontend_create(ptr(self.blocks), self.blocks.shape[0], max_ep, max_sel, level, int(perceptual))
        else:
            self.L.ref_frontend_create_mt.restype = C.c_void_p
            self.L.ref_frontend_create_mt.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32]
            self.h = self.L.ref_frontend_create_mt(ptr(self.blocks), self.blocks.shape[0], max_ep, max_sel, level, int(perceptual), threads)
        assert self.h

    def call(self, name, arg=0):
        r = self.L.ref_frontend_call(self.h, name.encode(), arg)
        assert r != -1, name
        return r

    def get(self, name, dtype=np.uint8):
        need = self.L.ref_frontend_get(self.h, name.encode(), None, 0)
        assert need != 2 ** 64 - 1, name
        buf = np.zeros(need, np.uint8)
        self.L.ref_frontend_get(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), need)
        return buf.view(dtype)

    def get_csr(self, name):
        return csr_blob_split(self.get(name, np.uint32))

    def backend_encoder_blocks(self, nbx, nby, endpoint_thresh=1.5, selector_thresh=1.25):
        """basisu_backend::create_encoder_blocks on the finished frontend (2D, one slice). Returns a dict of arrays + the CPU seconds."""
        n = nbx * nby
        ep, pred, sel = np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
        k_ep = int(self.get("endpoint_cluster_etc_params").size // 1) if False else 65536
        o2n, n2o = np.full(k_ep, 0xFFFFFFFF, np.uint32), np.full(k_ep, 0xFFFFFFFF, np.uint32)
        secs = C.c_double(0)
        f = self.L.ref_backend_create_encoder_blocks
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        assert f(self.h, nbx, nby, endpoint_thresh, selector_thresh, ep.ctypes.data, pred.ctypes.data, sel.ctypes.data, o2n.ctypes.data, n2o.ctypes.data, C.byref(secs)) == 1
        return {"endpoint_index": ep, "predictor": pred, "selector_index": sel, "endpoint_old_to_new": o2n[o2n != 0xFFFFFFFF],
                "selector_new_to_old": n2o[n2o != 0xFFFFFFFF], "seconds": secs.value}

    def backend_run(self, slices, endpoint_thresh=1.5, selector_thresh=1.25):
        """basisu_backend::encode on the finished frontend; slices = [(first_block, nbx, nby), ...]. Returns (bytes, seconds)."""
        sl = np.ascontiguousarray(np.asarray(slices, np.uint32).reshape(-1, 3))
        self.L.ref_backend_run.restype = C.c_uint32
        self.L.ref_backend_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        secs = C.c_double(0)
        n = self.L.ref_backend_run(self.h, sl.ctypes.data_as(C.c_void_p), sl.shape[0], endpoint_thresh, selector_thresh, C.byref(secs))
        return n, secs.value

    def backend_get(self, name, slice_index=0, dtype=np.uint8):
        self.L.ref_backend_get.restype = C.c_uint64
        self.L.ref_backend_get.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64]
        need = self.L.ref_backend_get(self.h, name.encode(), slice_index, None, 0)
        assert need != 2 ** 64 - 1, name
        buf = np.zeros(need, np.uint8)
        self.L.ref_backend_get(self.h, name.encode(), slice_index, buf.ctypes.data_as(C.c_void_p), need)
        return buf.view(dtype)

    def set_tex_type(self, tex_type):
        """basist::basis_texture_type of the frontend params (3 = video frames); before compress."""
        self.L.ref_frontend_set_tex_type.restype = None
        self.L.ref_frontend_set_tex_type.argtypes = [C.c_void_p, C.c_uint32]
        self.L.ref_frontend_set_tex_type(self.h, tex_type)

    def set_state(self, color5_inten, selectors16, block_endpoint, block_selector):
        """Overwrite the finished frontend state with arbitrary codebooks / assignments (backend fuzzing)."""
        f = self.L.ref_frontend_set_state
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        ep = np.ascontiguousarray(color5_inten, np.uint8).reshape(-1, 4)
        sel = np.ascontiguousarray(selectors16, np.uint8).reshape(-1, 16)
        be, bs = np.ascontiguousarray(block_endpoint, np.uint32), np.ascontiguousarray(block_selector, np.uint32)
        assert f(self.h, ep.shape[0], ep.ctypes.data, sel.shape[0], sel.ctypes.data, be.ctypes.data, bs.ctypes.data) == 1

    def reoptimize(self, new_block_endpoints, final_codebook, block_selector_indices=None):
        """basisu_frontend::reoptimize_remapped_endpoints on this frontend -> old_to_new (int32, one per endpoint cluster before the call)."""
        f = self.L.ref_frontend_reoptimize
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        nb = np.ascontiguousarray(new_block_endpoints, np.uint32)
        k = self.get("endpoint_cluster_etc_params").size // 16
        o2n = np.full(max(k, 1), -1, np.int32)
        sel = None if block_selector_indices is None else np.ascontiguousarray(block_selector_indices, np.uint32)
        assert f(self.h, nb.ctypes.data, nb.size, o2n.ctypes.data, int(final_codebook), None if sel is None else sel.ctypes.data) == 1
        return o2n

    def basis_file(self, tex_type=0, userdata0=0, userdata1=0, y_flipped=False, us_per_frame=0, key_values=()):
        """basisu_file::init on the last backend_run's output -> the .basis file bytes."""
        f = self.L.ref_basis_file
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
        n = len(key_values)
        keys = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in key_values])
        bufs = [np.frombuffer(bytes(v), np.uint8) if len(v) else np.zeros(1, np.uint8) for _, v in key_values]
        vals = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
        sizes = np.array([len(v) for _, v in key_values] + [0], np.uint32)
        args = (self.h, tex_type, userdata0, userdata1, int(y_flipped), us_per_frame, keys, vals, sizes.ctypes.data, n)
        need = f(*args, None, 0)
        assert need, "ref_basis_file failed"
        out = np.zeros(need, np.uint8)
        f(*args, out.ctypes.data, need)
        return out

    def close(self):
        if self.h:
            self.L.ref_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


# ----------------------------------------------------------------------------- UASTC

def uastc_test_blocks():
    """~1.2k source blocks covering every class encode_uastc distinguishes (uastc_enc.cpp:3135-3152) plus degenerate cases."""
    rng = np.random.default_rng(77)
    parts = [to_pixel_blocks(synth(64, 64, 1234)), to_pixel_blocks(uniform_random(32, 32, 42))]
    kod = REF_DIR / "test_files" / "kodim03.png"
    if kod.exists():
        parts.append(to_pixel_blocks(load_png(kod)[200:264, 300:364]))
    else:
        parts.append(to_pixel_blocks(synth(64, 64, 99)))
    g = synth(48, 48, 5).copy(); g[..., 1] = g[..., 0]; g[..., 2] = g[..., 0]
    parts.append(to_pixel_blocks(g))
    yy, xx = np.mgrid[0:48, 0:48]
    a = synth(48, 48, 11).copy(); a[..., 3] = np.clip(128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 0, 255).astype(np.uint8)
    parts.append(to_pixel_blocks(a))
    an = synth(32, 32, 9).copy(); an[..., 3] = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    parts.append(to_pixel_blocks(an))
    ga = g.copy(); ga[..., 3] = a[..., 3]
    parts.append(to_pixel_blocks(ga))
    parts.append(np.ascontiguousarray(rng.integers(0, 256, (48, 1, 1, 4), dtype=np.uint8).repeat(4, 1).repeat(4, 2)))  # solid, any alpha
    two = rng.integers(0, 256, (96, 2, 4), dtype=np.uint8)
    tb = two[np.arange(96)[:, None], rng.integers(0, 2, (96, 16))].reshape(96, 4, 4, 4)
    tb[:64, :, :, 3] = 255
    parts.append(np.ascontiguousarray(tb))
    ramp = np.zeros((32, 4, 4, 4), np.uint8)  # one-channel ramps and near-solid blocks (degenerate endpoint handling)
    for i in range(32):
        ramp[i, :, :, :3] = rng.integers(0, 256, 3)
        ramp[i, :, :, i % 3] = (np.arange(16).reshape(4, 4) * (1 + i % 5) + i) % 256
        ramp[i, :, :, 3] = 255
    parts.append(ramp)
    # piecewise blocks: 2 or 3 regions with unrelated colour (and alpha) ramps, so the multi-subset modes (2, 3, 4, 7, 9, 16) win somewhere
    yy4, xx4 = np.mgrid[0:4, 0:4]
    masks = [xx4 >= 2, yy4 >= 2, xx4 + yy4 >= 4, xx4 >= 1, yy4 >= 3, xx4 > yy4, (xx4 >= 1) + (xx4 >= 3), (yy4 >= 1) + (yy4 >= 2) * 1, (xx4 + yy4 >= 2) * 1 + (xx4 + yy4 >= 5)]
    pw = np.zeros((len(masks) * 24, 4, 4, 4), np.uint8)
    for i in range(pw.shape[0]):
        m = masks[i % len(masks)].astype(np.int64)
        variant = i // len(masks)
        for region in range(int(m.max()) + 1):
            base = rng.integers(0, 256, 4)
            slope = rng.integers(-12, 13, (2, 4))
            v = base + xx4[..., None] * slope[0] + yy4[..., None] * slope[1] + rng.integers(-2, 3, (4, 4, 4))
            pw[i][m == region] = np.clip(v, 0, 255).astype(np.uint8)[m == region]
        if variant % 3 == 0:
            pw[i, :, :, 3] = 255
        elif variant % 3 == 1:
            pw[i, :, :, 1] = pw[i, :, :, 0]; pw[i, :, :, 2] = pw[i, :, :, 0]  # luminance + alpha
    parts.append(pw)
    return np.ascontiguousarray(np.concatenate(parts))


def uastc_flag_sets():
    """(name, pack flags): every level plus each option flag (uastc_enc.h:24-66) on top of the default level."""
    sets = [(f"level{l}", l) for l in range(5)]
    sets += [("favor_uastc", 2 | 8), ("favor_bc7", 2 | 16), ("etc1_faster", 2 | 64), ("etc1_fastest", 2 | 128),
             ("etc1_noflip", 2 | 256), ("favor_simpler", 2 | 512), ("level1_faster_simpler", 1 | 64 | 512), ("level3_favor_bc7", 3 | 16)]
    return sets


def ref_encode_uastc(blocks, flags):
    blocks = np.ascontiguousarray(blocks)
    n = blocks.shape[0]
    out = np.zeros((n, 16), np.uint8)
    ref().ref_encode_uastc(ptr(blocks), n, flags, ptr(out))
    return out


RDO_DEFAULTS = dict(lam=1.0, max_rms_ratio=10.0, skip_rms=8.0, smooth_std_dev=18.0, smooth_scale=10.0, dict_size=4096, literal_cost=100, refine=1)


def rdo_param_arrays(**kw):
    """uastc_rdo_params (uastc_enc.h:94-134) as the (float[5], uint32[3]) pair the harnesses take."""
    p = dict(RDO_DEFAULTS)
    p.update(kw)
    return (np.array([p["lam"], p["max_rms_ratio"], p["skip_rms"], p["smooth_std_dev"], p["smooth_scale"]], np.float32),
            np.array([p["dict_size"], p["literal_cost"], p["refine"]], np.uint32))


def uastc_rdo_cases():
    """(name, pack flags, total_jobs, uastc_rdo_params overrides) of the committed RDO known answers (tests/golden/uastc_rdo_vectors.npz)."""
    return [("default_l2", 2, 0, dict(lam=1.0)),
            ("jobs4_l2", 2, 4, dict(lam=1.0)),
            ("strong_l2", 2, 0, dict(lam=4.0)),
            ("norefine_l0", 0, 3, dict(lam=3.0, refine=0)),
            ("bigdict_l1", 1, 0, dict(lam=2.0, dict_size=32768)),
            ("tinydict_l2", 2, 5, dict(lam=10.0, dict_size=64, skip_rms=30.0)),
            ("settle_l0", 0, 0, dict(lam=20.0)),
            ("tuned_l3", 3, 2, dict(lam=1.5, max_rms_ratio=1.5, smooth_std_dev=40.0, smooth_scale=3.0, literal_cost=150))]


def synth_smooth(w, h, seed):
    """Low-noise RGBA image: blocks that UASTC encodes in mode 0 and that RDO modifies and refits a lot."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 4), np.float32)
    img[..., 0] = 128 + 100 * np.sin(x / 17) * np.cos(y / 23)
    img[..., 1] = 128 + 90 * np.sin(x / 29 + 1) * np.cos(y / 13)
    img[..., 2] = 128 + 80 * np.cos(x / 11) * np.sin(y / 19 + 2)
    img[..., 3] = 255
    img[..., :3] += rng.normal(0, 2.0, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def smooth_with_la_blocks(w, h, seed):
    """synth_smooth as blocks, every 4th block luminance+alpha: modes 15/17 next to modified mode-0 blocks, the one arrangement in which
    the selector field of a block overlaps the (refitted) endpoint bits of its neighbours (uastc_rdo.h, deferred write-back)."""
    b = to_pixel_blocks(synth_smooth(w, h, seed)).copy()
    grey = b[::4, :, :, 1].copy()
    b[::4, :, :, 0] = grey
    b[::4, :, :, 2] = grey
    b[::4, :, :, 3] = 255 - grey // 2
    return b


def uastc_rdo_test_blocks():
    """Source blocks of the RDO vectors: a smooth+noise image (neighbouring blocks resemble each other, which is what RDO feeds on), part
    of it with a smooth alpha channel, followed by the every-class block set of the encoder vectors."""
    b = to_pixel_blocks(synth(256, 96, 11)).copy()
    b[1024:, :, :, 3] = b[1024:, :, :, 0] // 2 + 60
    return np.ascontiguousarray(np.concatenate([b, smooth_with_la_blocks(192, 64, 3), uastc_test_blocks()[::3]]))


def ref_uastc_rdo(packed, blocks, flags, total_jobs=0, **kw):
    fp, up = rdo_param_arrays(**kw)
    out = np.ascontiguousarray(packed).copy()
    blocks = np.ascontiguousarray(blocks)
    assert ref().ref_uastc_rdo(ptr(out), ptr(blocks), out.shape[0], ptr(fp, f32p), ptr(up, u32p), flags, total_jobs) == 1
    return out


def host_uastc_rdo(packed, blocks, flags, total_jobs=0, table_trials=False, **kw):
    """The host build of uastc_rdo.h under a scalar strip loop; table_trials scores trials from the per-block error table like the GPU kernel."""
    fp, up = rdo_param_arrays(**kw)
    total_jobs |= 0x80000000 if table_trials else 0
    out = np.ascontiguousarray(packed).copy()
    blocks = np.ascontiguousarray(blocks)
    assert uastc_host().hc_uastc_rdo(ptr(out), ptr(blocks), out.shape[0], ptr(fp, f32p), ptr(up, u32p), flags, total_jobs) == 1
    return out


_uastc_host = None


def uastc_host():
    """The UASTC device core compiled for the host (tests/native/uastc_host.cpp): a checker for the CPU-only suite."""
    global _uastc_host
    if _uastc_host is None:
        d = ROOT / "tests" / "native"
        so, srcs = d / "libuastc_host.so", [d / "uastc_host.cpp", ROOT / "basis_universal_amd" / "csrc" / "uastc_core.h",
                                            ROOT / "basis_universal_amd" / "csrc" / "uastc_rdo.h",
                                            ROOT / "basis_universal_amd" / "csrc" / "uastc_tables.inc"]
        if not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so), str(srcs[0])])
        L = C.CDLL(str(so))
        L.hc_encode_uastc.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.hc_uastc_rdo.restype = C.c_int
        L.hc_uastc_rdo.argtypes = [u8p, u8p, C.c_uint32, f32p, u32p, C.c_uint32, C.c_uint32]
        L.hc_rehint.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.hc_unpack_block.argtypes = [u8p, u8p]
        L.hc_decode_uastc.restype = C.c_int
        L.hc_decode_uastc.argtypes = [u8p, C.c_uint32, u8p]
        L.hc_cell_compress.restype = C.c_uint64
        L.hc_cell_compress.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, u8p]
        L.hc_cell_estimate.restype = C.c_uint64
        L.hc_cell_estimate.argtypes = [C.c_uint32, C.c_uint32, u8p, C.c_uint32]
        L.hc_weight_of.restype = C.c_uint32
        L.hc_weight_of.argtypes = [C.c_uint32, C.c_uint32]
        L.hc_weight_table.restype = C.c_uint32
        L.hc_weight_table.argtypes = [C.c_uint32, C.c_uint32]
        _uastc_host = L
    return _uastc_host


_fsum_host = None


def fsum_host():
    """csrc/fsum_scan.h (the order-preserving float sum of the wide TSVQ kernels) compiled for the host: tests/native/fsum_host.cpp."""
    global _fsum_host
    if _fsum_host is None:
        d = ROOT / "tests" / "native"
        so, srcs = d / "libfsum_host.so", [d / "fsum_host.cpp", ROOT / "basis_universal_amd" / "csrc" / "fsum_scan.h"]
        if not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so), str(srcs[0])])
        L = C.CDLL(str(so))
        L.fsum_sequential.restype = C.c_float
        L.fsum_sequential.argtypes = [f32p, C.c_uint64, C.c_float]
        L.fsum_blocked.restype = C.c_float
        L.fsum_blocked.argtypes = [f32p, C.c_uint64, C.c_float, C.c_uint32, u64p]
        L.fsum_compose_check.restype = C.c_int
        L.fsum_compose_check.argtypes = [f32p, C.c_uint64, C.c_int, C.c_int, C.c_uint32]
        _fsum_host = L
    return _fsum_host


_tt_exact_host = None


def tt_exact_host():
    """csrc/tt_exact.h (when a block of the reference's double accumulators can be taken in one step: tsvq_wide6_kernels.hip, tt_walk) compiled for the host:
    tests/native/tt_exact_host.cpp."""
    global _tt_exact_host
    if _tt_exact_host is None:
        d = ROOT / "tests" / "native"
        so, srcs = d / "libtt_exact_host.so", [d / "tt_exact_host.cpp", ROOT / "basis_universal_amd" / "csrc" / "tt_exact.h"]
        if not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so), str(srcs[0])])
        L = C.CDLL(str(so))
        L.tt_sequential.restype = C.c_double
        L.tt_sequential.argtypes = [f32p, C.c_uint64]
        L.tt_blocked.restype = C.c_double
        L.tt_blocked.argtypes = [f32p, C.c_uint64, C.c_uint32, u64p]
        L.tt_low_bit.restype = C.c_int
        L.tt_low_bit.argtypes = [C.c_double]
        _tt_exact_host = L
    return _tt_exact_host


def host_encode_uastc(blocks, flags):
    blocks = np.ascontiguousarray(blocks)
    n = blocks.shape[0]
    out = np.zeros((n, 16), np.uint8)
    uastc_host().hc_encode_uastc(ptr(blocks), n, flags, ptr(out))
    return out


def host_decode_uastc(packed, nbx=None, nby=None):
    """UASTC LDR 4x4 blocks (n, 16) -> texels: (n, 4, 4, 4) u8, or the (nby * 4, nbx * 4, 4) raster when the block grid is given. The host build of the
    product's own uastc_core.h (unpack + per-texel interpolation); pinned to the reference's unpack_uastc in tests/test_uastc_core_host.py."""
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1, 16)
    out = np.zeros((packed.shape[0], 4, 4, 4), np.uint8)
    assert uastc_host().hc_decode_uastc(ptr(packed), packed.shape[0], ptr(out)) == 1
    return out if nbx is None else blocks_to_raster(out, nbx, nby)


def blocks_to_raster(texels, nbx, nby):
    """(n, 4, 4, C) block-raster texels -> (nby * 4, nbx * 4, C)"""
    c = texels.shape[-1]
    return texels.reshape(nby, nbx, 4, 4, c).transpose(0, 2, 1, 3, 4).reshape(nby * 4, nbx * 4, c)


def ref_decode_uastc(packed):
    """the reference's unpack_uastc (transcoder/basisu_transcoder.cpp:15743) per block -> (n, 4, 4, 4) u8"""
    L = ref()
    L.ref_unpack_uastc.restype = C.c_int
    L.ref_unpack_uastc.argtypes = [u8p, u8p]
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1, 16)
    out = np.zeros((packed.shape[0], 4, 4, 4), np.uint8)
    for i in range(packed.shape[0]):
        assert L.ref_unpack_uastc(ptr(packed[i]), ptr(out[i])) == 1
    return out


def decode_etc1s_blocks(blocks, nbx, nby):
    """ETC1S blocks (n, 8) u8 in block-raster order -> (nby * 4, nbx * 4, 3) u8: differential mode with zero deltas, both sub-blocks share
    colour5 and the intensity table (etc_block::unpack_color5 / get_block_colors, etc.h:543-570), selector bit planes per etc.h:232-236."""
    b = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 8)
    v = b.astype(np.uint64)
    word = np.zeros(b.shape[0], np.uint64)
    for i in range(8):
        word = (word << np.uint64(8)) | v[:, i]
    r5, g5, b5 = ((word >> np.uint64(s)) & np.uint64(31) for s in (59, 51, 43))
    inten = ((word >> np.uint64(37)) & np.uint64(7)).astype(np.int64)
    lo = (word & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    table = np.array([[-8, -2, 2, 8], [-17, -5, 5, 17], [-29, -9, 9, 29], [-42, -13, 13, 42], [-60, -18, 18, 60], [-80, -24, 24, 80], [-106, -33, 33, 106],
                      [-183, -47, 47, 183]], np.int64)
    base = np.stack([(c.astype(np.int64) << 3) | (c.astype(np.int64) >> 2) for c in (r5, g5, b5)], axis=1)      # (n, 3)
    to_sel = np.array([2, 3, 1, 0], np.int64)
    out = np.zeros((b.shape[0], 4, 4, 3), np.uint8)
    for y in range(4):
        for x in range(4):
            bit = np.uint64(x * 4 + y)
            raw = ((lo >> bit) & np.uint64(1)) | (((lo >> (np.uint64(16) + bit)) & np.uint64(1)) << np.uint64(1))
            sel = to_sel[raw.astype(np.int64)]
            out[:, y, x, :] = np.clip(base + table[inten, sel][:, None], 0, 255).astype(np.uint8)
    return out.reshape(nby, nbx, 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(nby * 4, nbx * 4, 3)


def decode_backend_blocks(encoder_blocks, endpoint_color5_inten, selector_blocks, nbx, nby):
    """What a transcoder to ETC1 decodes from the backend's output (the texture basis_compressor computes its m_basis_* stats on, comp.cpp:4194-4221):
    per block the endpoint / selector codebook entries the backend finally coded (its RDO may have replaced the frontend's choice). encoder_blocks:
    the (n, 4) u32 rows of Etc1sBackend.get("encoder_blocks") = (endpoint index, predictor, selector index, history index + 1), indices in the
    frontend's numbering; endpoint_color5_inten (k, 4) u8; selector_blocks (m, 8) u8 (selector bit planes in bytes 4-7)."""
    eb = np.ascontiguousarray(encoder_blocks).view(np.uint32).reshape(-1, 4)
    ep = np.ascontiguousarray(endpoint_color5_inten, np.uint8).reshape(-1, 4)[eb[:, 0]]
    sb = np.ascontiguousarray(selector_blocks, np.uint8).reshape(-1, 8)[eb[:, 2]]
    blk = np.zeros((eb.shape[0], 8), np.uint8)
    blk[:, 0:3] = ep[:, 0:3] << 3
    blk[:, 3] = (ep[:, 3] << 5) | (ep[:, 3] << 2) | 3    # both intensity tables, differential + flip bits (etc_block::is_etc1s)
    blk[:, 4:8] = sb[:, 4:8]
    return decode_etc1s_blocks(blk, nbx, nby)


def decode_backend_output(fe, be, nbx, nby):
    """decode_backend_blocks for an Etc1sFrontend + Etc1sBackend pair (one slice covering all blocks)"""
    prm = fe.get("endpoint_cluster_etc_params").reshape(-1, 16)[:, :4]
    return decode_backend_blocks(be.get("encoder_blocks"), prm, fe.get("optimized_cluster_selectors"), nbx, nby)


def psnr(a, b):
    """image_metrics::calc's PSNR over the given channels (enc.cpp:2155-2226): 20 log10(255 / rms), clamped to 100"""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return 100.0 if mse == 0 else min(100.0, 20.0 * np.log10(255.0 / np.sqrt(mse)))


def save_png(path, img):
    """Minimal 8-bit RGBA PNG writer (test inputs for the reference CLI)."""
    import struct, zlib
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    pathlib.Path(path).write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def have_ref_cli():
    return (ORACLE_DIR / "_ref" / "basisu").exists()


def run_ref_cli(png_path, *args, ktx2=False):
    """The reference command line tool (oracle/_ref/basisu, built from /root/reference) on one PNG (or a list of PNGs: array layers, cubemap
    faces, video frames of ONE output file) -> the bytes of the .basis (or .ktx2) it writes."""
    import subprocess, tempfile, shutil
    paths = list(png_path) if isinstance(png_path, (list, tuple)) else [png_path]
    with tempfile.TemporaryDirectory() as d:
        names = []
        for i, p in enumerate(paths):
            names.append("in.png" if len(paths) == 1 else f"in{i}.png")
            shutil.copy(p, pathlib.Path(d) / names[-1])
        r = subprocess.run([str(ORACLE_DIR / "_ref" / "basisu"), "-ktx2" if ktx2 else "-basis", "-no_multithreading", *args, *names], cwd=d, capture_output=True, text=True, timeout=600)
        outs = sorted(pathlib.Path(d).glob("*.ktx2" if ktx2 else "*.basis"))
        assert r.returncode == 0 and len(outs) == 1, r.stdout[-2000:] + r.stderr[-2000:]
        return np.fromfile(outs[0], np.uint8)


def ktx2_file_key_values(data):
    """The key-value pairs of a KTX2 file (kvdByteOffset/Length of the header), without the alignment dummy key the writer adds itself."""
    import struct
    raw = np.asarray(data, np.uint8).tobytes()
    ofs, size = struct.unpack_from("<II", raw, 56)
    out, pos = [], ofs
    while pos < ofs + size:
        n = struct.unpack_from("<I", raw, pos)[0]
        body = raw[pos + 4:pos + 4 + n]
        k = body[:body.index(b"\0")]
        if not (k and set(k) == {127}):
            out.append((k.decode(), body[len(k) + 1:]))
        pos = ofs + (pos + 4 + n - ofs + 3) // 4 * 4
    return out


def basis_file_key_values(data):
    """The key-value pairs stored in a .basis file (basis_file_header::m_extended_file_ofs/size)."""
    import struct
    raw = np.asarray(data, np.uint8).tobytes()
    ofs, size = struct.unpack_from("<II", raw, 69)
    if not size:
        return []
    kv, out, pos = raw[ofs:ofs + size], [], 8
    for _ in range(struct.unpack_from("<I", kv, 2)[0]):
        kl, vl = kv[pos], struct.unpack_from("<I", kv, pos + 1)[0]
        out.append((kv[pos + 5:pos + 5 + kl].decode(), kv[pos + 5 + kl:pos + 5 + kl + vl]))
        pos += 5 + kl + vl
    return out


_block_metric_host = None


def block_metric_host():
    """csrc/host/block_metric.h (the backend's inner loops, one variant per instruction set) compiled for the host: tests/native/block_metric_host.cpp."""
    global _block_metric_host
    if _block_metric_host is None:
        d = ROOT / "tests" / "native"
        so, srcs = d / "libblock_metric_host.so", [d / "block_metric_host.cpp", ROOT / "basis_universal_amd" / "csrc" / "host" / "block_metric.h"]
        if not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(srcs[0])])
        L = C.CDLL(str(so))
        L.bm_variants.restype = C.c_int
        L.bm_scan_check.restype = C.c_int
        L.bm_scan_check.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.bm_search_check.restype = C.c_int
        L.bm_search_check.argtypes = [C.c_uint64, C.c_int, C.c_int]
        _block_metric_host = L
    return _block_metric_host


# g_uastc_huff_modes (transcoder/basisu_transcoder.cpp:14409-14413): the mode of a UASTC block from the low 7 bits of its first byte
UASTC_HUFF_MODES = [11, 0, 10, 3, 11, 15, 12, 7, 11, 18, 10, 5, 11, 14, 12, 9, 11, 0, 10, 4, 11, 16, 12, 8, 11, 18, 10, 6, 11, 2, 12, 13, 11, 0, 10, 3, 11, 17, 12, 7, 11, 18, 10, 5, 11, 14, 12,
                    9, 11, 0, 10, 4, 11, 1, 12, 8, 11, 18, 10, 6, 11, 2, 12, 13, 11, 0, 10, 3, 11, 19, 12, 7, 11, 18, 10, 5, 11, 14, 12, 9, 11, 0, 10, 4, 11, 16, 12, 8, 11, 18, 10, 6, 11, 2,
                    12, 13, 11, 0, 10, 3, 11, 17, 12, 7, 11, 18, 10, 5, 11, 14, 12, 9, 11, 0, 10, 4, 11, 1, 12, 8, 11, 18, 10, 6, 11, 2, 12, 13]
