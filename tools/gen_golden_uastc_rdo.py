#!/usr/bin/env python3
"""Generate tests/golden/uastc_rdo_vectors.npz: known answers of the REAL reference's uastc_rdo (oracle/_ref/libref_harness.so, built from
/root/reference) -- source blocks, the reference's encode_uastc output per pack level, and the RDO output for every case of
helpers.uastc_rdo_cases() (lambda / dictionary size / strips / refinement / thresholds). Committed so that the HIP path and the host build
of the same core can be checked anywhere without the reference. Run in the build container."""
import pathlib, sys
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from helpers import *

blocks = uastc_rdo_test_blocks()
out = {"blocks": blocks}
for name, flags, jobs, kw in uastc_rdo_cases():
    key = f"packed_l{flags & 7}"
    if key not in out:
        out[key] = ref_encode_uastc(blocks, flags)
    out[name] = ref_uastc_rdo(out[key], blocks, flags, jobs, **kw)
    print(name, "modified", int((out[name] != out[key]).any(1).sum()), "of", blocks.shape[0])
np.savez_compressed(root / "tests" / "golden" / "uastc_rdo_vectors.npz", **out)
