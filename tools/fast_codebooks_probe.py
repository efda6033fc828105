"""Development probe for the fast codebook mode (row f3): size / PSNR of the bit-exact path and of the k-means path (both codebooks, or one
of them: BU_FAST_ENDPOINTS_ONLY / BU_FAST_SELECTORS_ONLY) at several iteration counts (BU_FAST_ITERS). Run on a GPU box."""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT))
import numpy as np
import test_gpu_fast_codebooks as T
from basis_universal_amd import capi
ctx = capi.Context(0)
cases = sys.argv[1:] or ["kodim03_q128", "synth1024_q128"]
for case in cases:
    img_fn, q = T.CASES[case]; img = img_fn()
    ex = T._encode(ctx, img, q, False)
    print(case, "exact", ex["size"], round(ex["psnr"], 3), ex["clusters"], flush=True)
    for env in ({"BU_FAST_SELECTORS_ONLY": "1", "BU_FAST_ITERS": "4"}, {"BU_FAST_SELECTORS_ONLY": "1", "BU_FAST_ITERS": "8"}, {"BU_FAST_SELECTORS_ONLY": "1", "BU_FAST_ITERS": "16"},
                {"BU_FAST_SELECTORS_ONLY": "1", "BU_FAST_ITERS": "32"}, {"BU_FAST_ITERS": "8"}, {"BU_FAST_ITERS": "16"}):
        for k in ("BU_FAST_ENDPOINTS_ONLY", "BU_FAST_SELECTORS_ONLY", "BU_FAST_ITERS"): os.environ.pop(k, None)
        os.environ.update(env)
        f = T._encode(ctx, img, q, True)
        print("   fast", env, f["size"], round(f["psnr"], 3), f["clusters"], flush=True)
