#!/usr/bin/env python3
"""Generate tests/golden/uastc_reference_vectors.npz: known-answer UASTC blocks produced by the REAL reference's encode_uastc
(oracle/_ref/libref_harness.so, built from /root/reference) for a fixed set of source blocks covering every block class
(opaque colour, grey, alpha, grey+alpha, solid, two-colour, noise, real image), at every pack level and for the option flags.
Committed so the HIP path (and the host build of the same core) can be checked anywhere without the reference. Run in the build container."""
import pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from helpers import *

blocks = uastc_test_blocks()
out = {"blocks": blocks}
for name, flags in uastc_flag_sets():
    out[name] = ref_encode_uastc(blocks if flags & 7 < 4 else blocks[::4], flags)
np.savez_compressed(root / "tests" / "golden" / "uastc_reference_vectors.npz", **out)
print("blocks", blocks.shape[0], "arrays", len(out), {k: v.shape for k, v in out.items()})
