"""UASTC + RDO through bu_hip_uastc_pipeline_* on the Kodak batch: ms per batch for combinations of lanes and reserved walk CUs (bu_hip_tuning::uastc_walk_cus).
    python tools/rdo_lanes.py [steps] [lanes,lanes,...] [cus,cus,...]"""
import sys, time, pathlib, hashlib, json
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import torch
import helpers
from basis_universal_amd import capi, uastc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lanes_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3").split(",")]
cus_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,32,64").split(",")]
z = np.load(root / "tests" / "golden" / "kodak24.npz")
names = sorted(z.files)
images = [np.concatenate([z[k], np.full(z[k].shape[:2] + (1,), 255, np.uint8)], axis=2) for k in names]
golden = json.loads((root / "tests" / "golden" / "kodak24_digests.json").read_text())["images"]
parts = [helpers.to_pixel_blocks(im) for im in images]
ofs = np.cumsum([0] + [p.shape[0] for p in parts])
blocks = np.concatenate(parts)
n = blocks.shape[0]
ctx = capi.Context(0)
ctx.check(ctx.lib.set_stream(ctx.h, torch.cuda.current_stream().cuda_stream), "set_stream")
d_px = torch.from_numpy(blocks.reshape(n, 64)).cuda()
params = uastc.RdoParams(m_lambda=1.0)
flags, jobs = uastc.LEVEL_DEFAULT, 4 * 24
for cus in cus_list:
    for lanes in lanes_list:
        ctx.set_tuning(uastc_walk_cus=cus)
        pipe = uastc.UastcPipeline(ctx, lanes, n, flags, jobs)
        outs = [torch.empty((n, 16), dtype=torch.uint8, device=d_px.device) for _ in range(lanes)]
        for k in range(lanes):
            pipe.submit(d_px.data_ptr(), n, outs[k].data_ptr(), params, flags, jobs)
        pipe.wait(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            pipe.submit(d_px.data_ptr(), n, outs[k % lanes].data_ptr(), params, flags, jobs)
        pipe.wait(0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        o = outs[0].cpu().numpy()
        same = sum(hashlib.sha256(o[ofs[i]:ofs[i + 1]].tobytes()).hexdigest() == golden[nm]["uastc_l2_rdo1_jobs4"] for i, nm in enumerate(names))
        alike = all(bool((x == outs[0]).all().item()) for x in outs[1:])
        pipe.close()
        print(f"walk CUs {cus:3d} lanes {lanes}: {dt * 1e3:6.2f} ms per batch = {n * 16 / 1e6 / dt:7.1f} Mpix/s, {same}/24 images identical to the reference, lanes alike {alike}", flush=True)
ctx.set_tuning()
