"""Mip generation, the host half on the CPU: the resampling plan bu_mipmap_plan builds (contributor lists of both axes,
pass order, sRGB tables = what Resampler / image_resample decide before touching pixels) is applied here by a float32 numpy emulation of
mipmap_kernels.hip -- same operations, same order -- and the result has to equal the REAL image_resample (oracle/_ref) byte for byte.
The GPU test (tests/test_gpu_mipmap.py) then only has to show that the kernels do what the emulation does."""
import ctypes as C

import numpy as np
import pytest

from helpers import have_ref, ref, synth, uniform_random

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")


def plan(sw, sh, dw, dh, srgb, flt, scale, wrap):
    from basis_universal_amd.etc1s import load_frontend_library
    L = load_frontend_library()
    f = L.bu_mipmap_plan
    f.restype = C.c_int
    f.argtypes = [C.c_uint32] * 4 + [C.c_int, C.c_char_p, C.c_float, C.c_int] + [C.c_void_p] * 9
    counts = np.zeros(4, np.uint32)
    assert f(sw, sh, dw, dh, int(srgb), flt.encode(), scale, int(wrap), counts.ctypes.data, *([None] * 8))
    xf, xp, xw = np.zeros(dw + 1, np.uint32), np.zeros(counts[0], np.uint16), np.zeros(counts[0], np.float32)
    yf, yp, yw = np.zeros(dh + 1, np.uint32), np.zeros(counts[1], np.uint16), np.zeros(counts[1], np.float32)
    t0, t1 = np.zeros(256, np.float32), np.zeros(8192, np.uint8)
    assert f(sw, sh, dw, dh, int(srgb), flt.encode(), scale, int(wrap), counts.ctypes.data, *[a.ctypes.data for a in (xf, xp, xw, yf, yp, yw, t0, t1)])
    return dict(x=(xf, xp, xw), y=(yf, yp, yw), x_after_y=bool(counts[2]), to_linear=t0, to_srgb=t1)


def emulate(src, dw, dh, p, srgb, num_comps):
    """mipmap_kernels.hip in numpy float32."""
    sh, sw = src.shape[:2]
    lin = np.empty((sh, sw, 4), np.float32)
    lin[..., :3] = p["to_linear"][src[..., :3]]
    lin[..., 3] = src[..., 3].astype(np.float32) * np.float32(1.0 / 255.0)

    def along(axis_taps, data, n_out, axis, start_from_zero):
        first, pixel, weight = axis_taps
        shape = list(data.shape); shape[axis] = n_out
        out = np.zeros(shape, np.float32)
        for i in range(n_out):
            acc = None
            for k in range(first[i], first[i + 1]):
                term = (data[:, pixel[k]] if axis == 1 else data[pixel[k]]) * weight[k]
                acc = (np.zeros_like(term) + term if start_from_zero else term) if acc is None else acc + term
            if axis == 1:
                out[:, i] = acc
            else:
                out[i] = acc
        return out
    if not p["x_after_y"]:
        res = along(p["y"], along(p["x"], lin, dw, 1, True), dh, 0, False)
    else:
        res = along(p["x"], along(p["y"], lin, dh, 0, False), dw, 1, True)
    res = np.clip(res, np.float32(0), np.float32(1))
    out = np.zeros((dh, dw, 4), np.uint8)
    out[..., 3] = 255
    for c in range(num_comps):
        if srgb and c < 3:
            j = (np.float32(8191.0) * res[..., c] + np.float32(.5)).astype(np.int32)
            out[..., c] = p["to_srgb"][np.clip(j, 0, 8191)]
        else:
            j = (np.float32(255.0) * res[..., c] + np.float32(.5)).astype(np.int32)
            out[..., c] = np.clip(j, 0, 255)
    return out


def reference(src, dw, dh, srgb, flt, scale, wrap, num_comps):
    R = ref()
    R.ref_image_resample.restype = C.c_int
    R.ref_image_resample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_float, C.c_int, C.c_uint32, C.c_uint32]
    src = np.ascontiguousarray(src)
    out = np.zeros((dh, dw, 4), np.uint8)
    assert R.ref_image_resample(src.ctypes.data, src.shape[1], src.shape[0], out.ctypes.data, dw, dh, int(srgb), flt.encode(), scale, int(wrap), 0, num_comps)
    return out


def rgba(w, h, seed, noise=False):
    img = uniform_random(w, h, seed) if noise else synth((w + 3) // 4 * 4, (h + 3) // 4 * 4, seed)[:h, :w].copy()
    yy, xx = np.mgrid[0:h, 0:w]
    img[..., 3] = np.clip(128 + 120 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(img)


CASES = [  # src w, h, dst w, h, srgb, filter, scale, wrap, comps
    (64, 48, 32, 24, True, "kaiser", 1.0, True, 3),       # the compressor's defaults
    (64, 48, 32, 24, True, "kaiser", 1.0, True, 4),
    (64, 48, 32, 24, False, "kaiser", 1.0, False, 4),
    (33, 19, 16, 9, True, "kaiser", 1.0, True, 4),        # odd sizes
    (7, 5, 3, 2, True, "kaiser", 1.0, True, 3),
    (2, 2, 1, 1, True, "kaiser", 1.0, True, 4),
    (3, 1, 1, 1, True, "kaiser", 1.0, False, 4),          # one axis already at 1: magnification branch of make_clist
    (1, 4, 1, 2, True, "kaiser", 1.0, True, 3),
    (96, 16, 48, 8, True, "box", 1.0, True, 4),
    (96, 16, 48, 8, True, "tent", 1.0, False, 3),
    (40, 40, 20, 20, True, "lanczos4", 1.0, True, 4),
    (40, 40, 20, 20, False, "mitchell", 1.0, False, 3),
    (40, 40, 20, 20, True, "blackman", 1.0, True, 3),
    (40, 40, 20, 20, True, "lanczos12", 1.0, True, 3),
    (40, 40, 20, 20, True, "bell", 1.0, True, 3),
    (40, 40, 20, 20, True, "catmullrom", 1.0, False, 4),
    (48, 32, 24, 16, True, "kaiser", 1.5, True, 3),       # -mip_scale
    (48, 32, 24, 16, True, "kaiser", 0.75, False, 4),
    (160, 8, 80, 4, True, "kaiser", 1.0, True, 4),        # very different axis costs: the other pass order
    (8, 160, 4, 80, True, "kaiser", 1.0, True, 4),
    (16, 64, 16, 32, True, "kaiser", 1.0, True, 4),       # only y shrinks: x is resampled after y (Resampler's delayed x pass)
    (24, 96, 20, 32, False, "lanczos3", 1.0, False, 3),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_plan_applied_in_float32_matches_image_resample(case):
    sw, sh, dw, dh, srgb, flt, scale, wrap, comps = case
    for seed, noise in ((1, False), (2, True)):
        src = rgba(sw, sh, seed, noise)
        got = emulate(src, dw, dh, plan(sw, sh, dw, dh, srgb, flt, scale, wrap), srgb, comps)
        exp = reference(src, dw, dh, srgb, flt, scale, wrap, comps)
        assert (got == exp).all(), (np.argwhere(got != exp)[:5], got[got != exp][:5], exp[got != exp][:5])


def test_both_pass_orders_occur():
    orders = {plan(sw, sh, dw, dh, True, "kaiser", 1.0, True)["x_after_y"] for sw, sh, dw, dh in [(16, 64, 16, 32), (1, 4, 1, 2), (64, 48, 32, 24)]}
    assert orders == {True, False}


def test_mip_chain_sizes():
    from basis_universal_amd.etc1s import load_frontend_library
    L = load_frontend_library()
    L.bu_mipmap_level_sizes.restype = C.c_uint32
    L.bu_mipmap_level_sizes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    out = np.zeros(64, np.uint32)
    n = L.bu_mipmap_level_sizes(130, 67, 1, out.ctypes.data, 32)
    assert out[:2 * n].reshape(-1, 2).tolist() == [[65, 33], [32, 16], [16, 8], [8, 4], [4, 2], [2, 1], [1, 1]]
    assert L.bu_mipmap_level_sizes(1, 1, 1, out.ctypes.data, 32) == 0
    assert L.bu_mipmap_level_sizes(256, 256, 16, out.ctypes.data, 32) == 4
