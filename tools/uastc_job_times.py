import sys, pathlib
import numpy as np
ROOT = pathlib.Path("/root/repo")
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
