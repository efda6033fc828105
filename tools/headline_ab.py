#!/usr/bin/env python3
"""The headline step (4096^2 q128 level 1, tiles resident, one image in flight on torch's current stream) with and without the per-kernel HIP-event regions that bench.py
records inside its timed steps -- what the instrumentation itself costs.   usage: python tools/headline_ab.py [steps]"""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
blocks = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
n = blocks.shape[0]
d = torch.from_numpy(blocks.reshape(n, 64)).to(dev)
ep, sel = quality_to_clusters(128, n)
ctx = capi.Context(0)
ctx.check(ctx.lib.set_stream(ctx.h, torch.cuda.current_stream().cuda_stream), "set_stream")
def step():
    fe = Etc1sFrontend(ctx); fe.init(d.data_ptr(), ep, sel, 1, True, n_blocks=n); fe.compress(); return fe
for _ in range(3): step().close()
out = {}
for rep in range(2):
    for prof in (True, False):
        ctx.profile_enable(prof)
        torch.cuda.synchronize(); t0 = time.perf_counter(); last = None
        for _ in range(steps):
            if last is not None: last.close()
            last = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        last.close()
        if prof: ctx.profile_read()
        out.setdefault("events_on" if prof else "events_off", []).append(round(dt * 1e3, 3))
ctx.profile_enable(False)
print(json.dumps({"ms_per_step": out, "steps": steps}))
