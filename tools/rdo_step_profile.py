#!/usr/bin/env python3
"""Where the serial step of k_rdo_strips spends its cycles: the Kodak batch (24 images, 96 strips) through the instrumented build of the library
(tools/build_rdo_profile.sh: -DRDO_PROFILE). Prints, for strip 0, the clock64() cycles accumulated by thread 0 between the RDO_TICK marks of the
step loop (uastc_rdo_kernels.hip), per step, with what each interval covers.   usage: python tools/rdo_step_profile.py > profiles/<name>.txt"""
import io, os, pathlib, re, sys, tempfile
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import helpers
from basis_universal_amd import capi, uastc

LABELS = {0: "step head: readlane of the prefetched info / block words, field extraction",
          8: "candidate: ring read, field extraction, history bucket load issued (HBM)",
          9: "candidate: table sum (16-32 LDS lookups)",
          10: "candidate: history bucket resolve (waits for the HBM probe)",
          1: "candidate: cost, lane minimum",
          3: "own pattern's history lookup, threshold against the block's own cost",
          4: "post winner (ds_min_u64), issue next step's prefetches",
          5: "LDS barrier (all candidates scored)",
          6: "thread 0: write-back of block / history entry / ring slot, next table to LDS",
          7: "closing barrier"}

lib = capi.HipLibrary(root / "tools" / "bin" / "libbasisu_hip_rdoprof.so")
ctx = capi.Context(0, lib=lib)
z = np.load(root / "tests" / "golden" / "kodak24.npz")
imgs = [np.concatenate([z[k], np.full(z[k].shape[:2] + (1,), 255, np.uint8)], axis=2) for k in sorted(z.files)]
blocks = np.concatenate([helpers.to_pixel_blocks(im) for im in imgs])
n = blocks.shape[0]
d_px = ctx.upload(blocks)
packed = uastc.encode_uastc_blocks(ctx, blocks, 2)
# stderr of the C library -> a file
fd = os.dup(2)
tmp = tempfile.TemporaryFile()
os.dup2(tmp.fileno(), 2)
d_blk = ctx.upload(packed)
_, info = uastc.uastc_rdo(ctx, d_blk, d_px, uastc.RdoParams(m_lambda=1.0), 2, 96, n_blocks=n)
os.dup2(fd, 2)
tmp.seek(0)
text = tmp.read().decode()
m = re.search(r"rdo strip 0 cycles by phase:((?: \d+)+)", text)
vals = [int(v) for v in m.group(1).split()]
steps = n // 96
total = sum(vals)
print(f"k_rdo_strips, Kodak batch ({n} blocks, 96 strips of {steps} steps), strip 0, thread 0: clock64() ticks per step by interval (RDO_PROFILE build)")
print(f"modified {info['modified']} of {n} blocks; ticks are s_memtime units (the shader clock while the kernel runs)")
for k, v in enumerate(vals):
    if v:
        print(f"  tick {k:2d}: {v / steps:8.1f} per step  {100.0 * v / total:5.1f} %   {LABELS.get(k, '')}")
print(f"  total   : {total / steps:8.1f} per step")
