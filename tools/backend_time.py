#!/usr/bin/env python3
"""Host-only timing of bu::etc1s_backend's sub-stages (no GPU): the reference frontend's state for a synthetic image (cached in /tmp), then our backend on it.
    tools/backend_time.py [size] [repeats]        BU_HOST_THREADS=1 serialises the three walks so each loop's own cost shows."""
import os, sys, pathlib, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from helpers import RefFrontend, synth, to_pixel_blocks
from basis_universal_amd.backend import Etc1sBackend
from basis_universal_amd.etc1s import quality_to_clusters

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cache = pathlib.Path(f"/tmp/bt/fe_{size}.npz")
if cache.exists():
    arrays = dict(np.load(cache))
else:
    blocks = to_pixel_blocks(synth(size, size, 1234))
    ep, sel = quality_to_clusters(128, blocks.shape[0])
    fe = RefFrontend(blocks, ep, sel, 1, True)
    t = time.time(); fe.call("compress"); print("reference frontend", round(time.time() - t, 1), "s", file=sys.stderr)
    prm = fe.get("endpoint_cluster_etc_params").reshape(-1, 16)[:, :4].copy()
    arrays = dict(source_blocks=blocks, output_blocks=fe.get("encoded_blocks"), block_endpoint_index=fe.get("block_endpoint_clusters_indices", np.uint32),
                  block_selector_index=fe.get("block_selector_cluster_index", np.uint32), endpoint_color5_inten=prm, selector_blocks=fe.get("optimized_cluster_selectors"))
    np.savez(cache, **arrays)
    fe.close()
nb = size // 4
best = None
for r in range(reps):
    be = Etc1sBackend.from_arrays(slices=[(0, nb, nb)], perceptual=True, endpoint_rdo_thresh=1.5, selector_rdo_thresh=1.25, compression_level=1, **arrays)
    t = time.time(); total = be.encode(); dt = time.time() - t
    st = dict(be.stage_times()); st["_total"] = dt; st["_bytes"] = total
    be.close()
    if best is None or dt < best["_total"]: best = st
for k, v in best.items():
    print(f"{k:28s} {v * 1000 if k != '_bytes' else v:10.2f}")
