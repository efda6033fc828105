#!/usr/bin/env python3
"""Wall time of every frontend stage (and the sub-stages the frontend reports, `~...`) for 8192^2 q255 level 1, tiles resident: where the host time of config #4's step goes."""
import pathlib, sys, time, json
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = int(sys.argv[2]) if len(sys.argv) > 2 else 255
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 5678 if size == 8192 else 1234))
n = blocks.shape[0]
d = torch.from_numpy(blocks.reshape(n, 64)).cuda()
ep, sel = quality_to_clusters(q, n)
ctx = capi.Context(0)
acc = {}
for it in range(4):
    fe = Etc1sFrontend(ctx); fe.init(d.data_ptr(), ep, sel, 1, True, n_blocks=n)
    torch.cuda.synchronize(); t0 = time.perf_counter(); fe.compress(); dt = time.perf_counter() - t0
    if it:
        for k, v in fe.stage_times(): acc[k] = acc.get(k, 0.0) + v / 3
        acc["TOTAL"] = acc.get("TOTAL", 0.0) + dt / 3
    fe.close()
print(json.dumps({k: round(v * 1e3, 2) for k, v in acc.items()}))
