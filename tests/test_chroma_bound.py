"""The refine sweep's pruning bound (etc1s_device.h: chroma_lower_bound, docs/HISTORY.md R6.17) as arithmetic, checked on the CPU: for any tile and any candidate colour the bound made
from the tile's chroma moments never exceeds the sum of the sixteen chroma terms of the reference's perceptual colour distance (encoder/basisu_enc.h:1141-1195). The kernel's results are held to the
reference by the GPU parity tests; this holds the inequality itself, on random tiles and on the extremes (saturated primaries, flat tiles, one outlier, candidates at the ends of the range)."""
import numpy as np


def cvec(rgb):
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    l = 14 * r + 45 * g + 5 * b
    return l, 64 * r - l, 64 * b - l      # etc1s_device.h: to_cvec<true>


def chroma_terms(y, z, cy, cz):
    dy, dz = y - cy, z - cz
    cr, cb = (dy * dy) >> 5, (dz * dz) >> 5
    return ((26 * cr) >> 7) + ((3 * cb) >> 7)          # etc1s_device.h: chroma_term


def lower_bound(y, z, cy, cz):
    """chroma_lower_bound with chroma_moments as k_refine_sorted makes them (y, z: [..., 16])"""
    s1y, s1z = y.sum(-1), z.sum(-1)
    my, mz = s1y >> 4, s1z >> 4                          # arithmetic shift = floor
    r1y, r1z = s1y - 16 * my, s1z - 16 * mz
    ry, rz = y - my[..., None], z - mz[..., None]
    r2y, r2z = ((ry * ry) >> 10).sum(-1), ((rz * rz) >> 10).sum(-1)
    ey, ez = my - cy, mz - cz
    A = ((ey * ey) >> 6) + r2y + ((ey * r1y) >> 9)
    B = ((ez * ez) >> 6) + r2z + ((ez * r1z) >> 9)
    assert (np.abs(ey) < 2 ** 23).all() and (np.abs(A) < 2 ** 24).all() and (np.abs(B) < 2 ** 24).all(), "24-bit multiplies on the device"
    return np.maximum(((A * 26 + B * 3) >> 2) - 36, 0)


def check(tiles, colours):
    _, y, z = cvec(tiles)                                # [n, 16]
    _, cy, cz = cvec(colours)                            # [n]
    exact = chroma_terms(y, z, cy[:, None], cz[:, None]).sum(-1)
    lb = lower_bound(y, z, cy, cz)
    assert (lb <= exact).all(), f"bound above the exact sum: {int((lb - exact).max())}"
    return lb, exact


def test_bound_never_exceeds_the_sum_random():
    rng = np.random.default_rng(1234)
    n = 400_000
    tiles = rng.integers(0, 256, (n, 16, 3), dtype=np.int64)
    colours = rng.integers(0, 256, (n, 3), dtype=np.int64)
    lb, exact = check(tiles, colours)
    # and it is worth something: on smooth tiles far from the candidate it is most of the sum
    base = rng.integers(0, 256, (n, 1, 3), dtype=np.int64)
    smooth = np.clip(base + rng.integers(-6, 7, (n, 16, 3)), 0, 255)
    lb, exact = check(smooth, colours)
    far = exact > 2000
    assert far.any() and (lb[far] >= 0.9 * exact[far] - 40).all()


def test_bound_on_extremes():
    corners = np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)], dtype=np.int64)
    tiles, colours = [], []
    for a in corners:
        for b in corners:
            for k in (0, 1, 8, 15, 16):                  # k texels of colour b in a tile of colour a
                t = np.repeat(a[None], 16, 0); t[:k] = b
                for c in corners:
                    tiles.append(t); colours.append(c)
    # candidates one step off the tile's mean, both sides (the bound's floors lose the most where the true sum is small)
    rng = np.random.default_rng(7)
    flat = rng.integers(0, 256, (2000, 1, 3), dtype=np.int64)
    for d in (-2, -1, 0, 1, 2):
        tiles.extend(np.repeat(flat, 16, 1)); colours.extend(np.clip(flat[:, 0] + d, 0, 255))
    check(np.array(tiles), np.array(colours))
