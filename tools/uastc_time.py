#!/usr/bin/env python3
"""Per-kernel time of the UASTC encoder for a few flag sets, on the bench image (synth 4096^2) and on the Kodak batch.   tools/uastc_time.py [steps]"""
import sys, pathlib, json
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import helpers
from basis_universal_amd import uastc
from basis_universal_amd import capi

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2:   # A/B runs: another build of libbasisu_hip.so (same box, same clocks: gpurun_ab/<variant>/libbasisu_hip.so)
    capi.LIB_PATH = pathlib.Path(sys.argv[2])
    capi.HipLibrary.__init__.__defaults__ = (capi.LIB_PATH,)
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
ctx = capi.Context(0)
z = np.load(ROOT / "tests" / "golden" / "kodak24.npz")
kodak = np.concatenate([helpers.to_pixel_blocks(np.concatenate([z[k], np.full(z[k].shape[:2] + (1,), 255, np.uint8)], axis=2)) for k in sorted(z.files)])
sets = {"synth4096": helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234)), "kodak24": kodak}
for name, blocks in sets.items():
    d = torch.from_numpy(np.ascontiguousarray(blocks)).cuda()
    n = blocks.shape[0]
    out = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    for label, flags in [("level2", uastc.LEVEL_DEFAULT), ("level2+faster", uastc.LEVEL_DEFAULT | uastc.ETC1_FASTER_HINTS), ("level2+fastest", uastc.LEVEL_DEFAULT | uastc.ETC1_FASTEST_HINTS),
                         ("level2+noflip", uastc.LEVEL_DEFAULT | uastc.ETC1_DISABLE_FLIP_AND_INDIVIDUAL), ("level1", uastc.LEVEL_FASTER), ("level3", uastc.LEVEL_SLOWER)]:
        if only and label not in only: continue
        uastc.encode_uastc_blocks(ctx, d.data_ptr(), flags, n_blocks=n, out_device=out.data_ptr())
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        for _ in range(steps):
            uastc.encode_uastc_blocks(ctx, d.data_ptr(), flags, n_blocks=n, out_device=out.data_ptr())
        torch.cuda.synchronize()
        k = ctx.profile_read(); ctx.profile_enable(False)
        print(name, n, label, {a: round(v[0] / steps, 3) for a, v in k.items()}, flush=True)
