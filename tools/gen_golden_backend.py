"""Golden digests of the REAL reference backend (oracle/_ref: basisu_frontend::compress + basisu_backend::encode, thresholds 1.5 / 1.25) for
the frontend cases of tests/test_gpu_etc1s_frontend.py -> tests/golden/etc1s_backend_digests.json. Run here (needs /root/reference built
into oracle/_ref); the GPU-side tests and __graft_entry__.smoke() compare the HIP frontend + host backend against these."""
import hashlib
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import helpers  # noqa: E402
import test_gpu_etc1s_frontend as T  # noqa: E402

PAYLOAD = ("endpoint_palette", "selector_palette", "slice_image_tables", "slice_image_data", "slice_image_crcs")


def digest(get):
    return {k: hashlib.sha256(np.ascontiguousarray(get(k)).tobytes()).hexdigest() for k in PAYLOAD}


def main():
    out = {}
    for case in sorted(T.CASES):
        blocks, max_ep, max_sel, level, perceptual = T._params(case)
        img = T.CASES[case][0]()
        nbx, nby = (img.shape[1] + 3) // 4, (img.shape[0] + 3) // 4
        fe = helpers.RefFrontend(blocks, max_ep, max_sel, level, perceptual)
        fe.call("compress")
        total, _ = fe.backend_run([(0, nbx, nby)], 1.5, 1.25)
        out[case] = {"slices": [[0, nbx, nby]], "level": level, "compressed_bytes": int(total), "digests": digest(fe.backend_get)}
        fe.close()
        print(case, total)
    (ROOT / "tests" / "golden" / "etc1s_backend_digests.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
