#!/usr/bin/env python3
"""One launch per mode job of the UASTC candidates kernel, timed with events (the library prints them under BU_UASTC_JOB_TIMES=1):
    gpurun -- 'BU_UASTC_JOB_TIMES=1 python tools/uastc_job_times.py 2>&1 | grep "uastc job"'"""
import sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, helpers
from basis_universal_amd import uastc, capi
ctx = capi.Context(0)
blocks = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
d = torch.from_numpy(np.ascontiguousarray(blocks)).cuda(); n = blocks.shape[0]
out = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
for _ in range(2):
    uastc.encode_uastc_blocks(ctx, d.data_ptr(), uastc.LEVEL_DEFAULT, n_blocks=n, out_device=out.data_ptr())
torch.cuda.synchronize()
