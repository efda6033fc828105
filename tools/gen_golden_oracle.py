#!/usr/bin/env python3
"""Generate tests/golden/etc1s_reference_vectors.npz: small known-answer vectors produced by the REAL reference
(oracle/_ref/libref_harness.so, built from /root/reference) for the functions the C oracle restates. Committed so that the oracle can
be re-pinned anywhere (CPU CI, GPU box) without the reference. Run in the build container."""
import pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from helpers import *

R = ref()
rng = np.random.default_rng(2024)
img = synth(64, 48, 31)
blocks = np.concatenate([to_pixel_blocks(img), to_pixel_blocks(uniform_random(16, 16, 8))])
k = REF_DIR / "test_files" / "kodim03.png"
blocks = np.concatenate([blocks, to_pixel_blocks(load_png(k)[200:232, 300:364])])  # a 64x32 crop of kodim03 (128 blocks)
n = blocks.shape[0]
out = {"blocks": blocks}
for level in (0, 1, 2, 6):
    for perc in (0, 1):
        o = np.zeros((n, 8), np.uint8)
        R.ref_encode_etc1s_blocks(ptr(blocks), n, level, perc, ptr(o))
        out[f"etc1s_l{level}_p{perc}"] = o
# cluster optimizer known answers
px_lists, res = [], []
for i, cnt in enumerate([8, 16, 40, 333, 5000]):
    base = rng.integers(0, 256, 3)
    px = np.clip(base[None] + rng.normal(0, 20, (cnt, 3)), 0, 255).astype(np.uint8)
    rgba = np.ascontiguousarray(np.concatenate([px, np.full((cnt, 1), 255, np.uint8)], axis=1))
    for q in (1, 2, 3):
        for perc in (0, 1):
            c = np.zeros(3, np.uint8); it = np.zeros(1, np.uint32); e = np.zeros(1, np.uint64)
            assert R.ref_etc1_optimize(ptr(rgba), cnt, q, perc, ptr(c), ptr(it, u32p), ptr(e, u64p), None) == 1
            res.append([i, q, perc, c[0], c[1], c[2], it[0], e[0]])
    out[f"cluster_px_{i}"] = rgba
out["cluster_results"] = np.array(res, np.uint64)
# a frontend run on these blocks: codebooks + final blocks
fe = RefFrontend(blocks, 48, 64, 1, True)
fe.call("compress")
for name in ("etc1_blocks", "endpoint_cluster_etc_params", "block_endpoint_clusters_indices", "encoded_blocks", "optimized_cluster_selectors", "block_selector_cluster_index"):
    out["fe_" + name] = fe.get(name)
fe.close()
np.savez_compressed(root / "tests" / "golden" / "etc1s_reference_vectors.npz", **out)
print("blocks", n, "arrays", len(out))
