#!/usr/bin/env python3
"""Generate tests/golden/etc1s_frontend_digests.json by running the REAL reference frontend (oracle/_ref/libref_harness.so, built
from /root/reference by oracle/Makefile) single-threaded on the seeded inputs of tests/test_gpu_etc1s_frontend.py.
Run in the build container (needs /root/reference); the output is committed so the GPU box can check parity without the reference."""
import json, pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from helpers import RefFrontend
import test_gpu_etc1s_frontend as T
import ctypes as C

def ref_quality_to_clusters(n_blocks):
    # pinned against the reference itself in tests/test_host_logic.py; here use our mirror (host-only library)
    from basis_universal_amd.etc1s import quality_to_clusters
    return quality_to_clusters(128, n_blocks)

out = {}
for case in sorted(T.CASES):
    img_fn, max_ep, max_sel, level, perceptual = T.CASES[case]
    blocks = T.to_pixel_blocks(img_fn())
    if max_ep is None:
        max_ep, max_sel = ref_quality_to_clusters(blocks.shape[0])
    fe = RefFrontend(blocks, max_ep, max_sel, level, perceptual)
    fe.call("compress")
    st = {k: fe.get(k) for k in T.STATE}
    out[case] = {"n_blocks": int(blocks.shape[0]), "max_endpoint_clusters": max_ep, "max_selector_clusters": max_sel, "level": level,
                 "perceptual": bool(perceptual), "digests": T._digest(st),
                 "final_endpoint_clusters": int(st["endpoint_clusters"].view(np.uint32)[0]),
                 "final_selector_clusters": int(st["selector_cluster_block_indices"].view(np.uint32)[0])}
    fe.close()
    print(case, out[case]["final_endpoint_clusters"], out[case]["final_selector_clusters"])
(root / "tests" / "golden" / "etc1s_frontend_digests.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
