#!/usr/bin/env python3
"""Where the one-workgroup split kernel (k_tsvq_split) spends thread 0's cycles, per tree of one frontend step, through the instrumented build of the library
(tools/build_tsvq_profile.sh: -DTQ_PROFILE; on the GPU box: cp tools/bin/libbasisu_hip_tqprof.so basis_universal_amd/lib/libbasisu_hip.so first).
usage: python tools/tsvq_split_profile.py [codebook threads] > profiles/<name>.txt"""
import ctypes as C, pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters

LABELS = ["prologue (node record, origin)", "covariance pass (pipelined chains)", "covariance renormalisation (thread 0)", "principal axis (one wave)",
          "side passes (classification + sums), all of them", "between side passes: centroids, variances, convergence (thread 0)", "partition into the children's lists",
          "  exact side pass: distance table + barrier", "  exact side pass: member loop (gather, classify, integer addends)", "  exact side pass: 32 wave reductions",
          "  exact side pass: block sums, three block-wide ORs"]
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = capi.Context(0)
dll = ctx.lib.dll
dll.tsvq_profile_read.argtypes = [C.c_void_p]
blocks = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
ep, sel = quality_to_clusters(128, blocks.shape[0])
buf = (C.c_ulonglong * 16)()
for rep in range(2):
    fe = Etc1sFrontend(ctx, max_threads=threads)
    fe.init(blocks, ep, sel, 1, True)
    L = fe.L
    for stage in ("init_etc1_images", "init_endpoint_training_vectors"):
        assert L.bu_frontend_call(fe.h, stage.encode(), 0)
    dll.tsvq_profile_read(buf)
    assert L.bu_frontend_call(fe.h, b"generate_endpoint_clusters", 0)
    dll.tsvq_profile_read(buf); prof_ep = list(buf)
    for stage, arg in (("generate_endpoint_codebook", 0), ("refine_endpoint_clusterization", 0), ("eliminate_redundant_or_empty_endpoint_clusters", 0),
                       ("generate_block_endpoint_clusters", 0), ("create_initial_packed_texture", 0)):
        assert L.bu_frontend_call(fe.h, stage.encode(), arg)
    dll.tsvq_profile_read(buf)
    assert L.bu_frontend_call(fe.h, b"generate_selector_clusters", 0)
    dll.tsvq_profile_read(buf); prof_sel = list(buf)
    fe.close()
for name, p in (("endpoint tree (6 floats)", prof_ep), ("selector tree (packed 16 x 2 bits)", prof_sel)):
    wgs, passes = p[15], p[14]
    tot = sum(p[:7]) + sum(p[7:11])
    print(f"{name}: {wgs} split workgroups, {passes} side passes ({passes / max(wgs, 1):.2f} per split); thread-0 clock64() ticks per workgroup (100 MHz constant clock? see total)")
    for k in range(11):
        print(f"  {LABELS[k]:90s} {p[k] / max(wgs, 1):10.1f}  {100.0 * p[k] / max(tot, 1):5.1f} %")
    print(f"  {'total':90s} {tot / max(wgs, 1):10.1f}")
ctx.close()
