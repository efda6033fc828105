#!/usr/bin/env python3
"""Times bu_hip_kmeans_codebook on the selector training set of the bench image (4096x4096 synthetic, q128): the distinct packed selector vectors as the
frontend would hand them over, HIP-event time of the whole call (label kmeans_selectors). BU_KM_DEBUG_SKIP=1/2/3 switches parts of k_km_assign off
(timing experiments only).   usage: python tools/scratch/km_time.py [size]"""
import ctypes as C, pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = capi.Context(0)
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234))
n = blocks.shape[0]
max_ep, max_sel = quality_to_clusters(128, n)
fe = Etc1sFrontend(ctx)
fe.init(blocks, max_ep, max_sel, 1, True)
for st in ("init_etc1_images", "init_endpoint_training_vectors", "generate_endpoint_clusters"):
    fe.call(st)
fe.call("generate_endpoint_codebook", 0)
for st in ("refine_endpoint_clusterization", "eliminate_redundant_or_empty_endpoint_clusters"):
    fe.call(st)
fe.call("generate_endpoint_codebook", 1)
for st in ("eliminate_redundant_or_empty_endpoint_clusters", "generate_block_endpoint_clusters", "create_initial_packed_texture"):
    fe.call(st)
enc = fe.get("orig_encoded_blocks").reshape(-1, 8) if False else fe.get("encoded_blocks").reshape(-1, 8)
d_enc = ctx.upload(enc)
d_w = ctx.alloc(n * 8); d_f = ctx.alloc(n * 64)
ctx.check(ctx.lib.k_selector_training_vectors(ctx.h, d_enc, n, 1, d_f, d_w), "training vectors")
d_sorted = ctx.alloc(n * 4); d_keys = ctx.alloc(n * 4); d_uw = ctx.alloc(n * 8); d_goffs = ctx.alloc((n + 1) * 4)
nu = C.c_uint32(0)
ctx.check(ctx.lib.k_unique_selector_vectors(ctx.h, d_enc, d_w, n, d_sorted, d_keys, d_uw, d_goffs, C.byref(nu)), "unique")
u = nu.value
d_cl = ctx.alloc(u * 4); d_par = ctx.alloc(u * 4)
oc, op = C.c_uint32(0), C.c_uint32(0)
for rep in range(3):
    ctx.profile_enable(True)
    ctx.check(ctx.lib.kmeans_codebook(ctx.h, 0, d_keys, d_uw, d_goffs, u, max_sel, 32, 4, d_cl, d_par, C.byref(oc), C.byref(op)), "kmeans")
    k = ctx.profile_read()
    ctx.profile_enable(False)
    print(f"rep {rep}: {u} distinct vectors -> {oc.value} clusters, {op.value} parents:", {a: round(b[0], 3) for a, b in k.items()}, flush=True)
