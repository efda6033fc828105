"""Time the UASTC phases on the GPU (HIP events per phase): python tools/uastc_time.py [size] [flags]"""
import sys, time, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import helpers
from basis_universal_amd import capi, uastc
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = capi.Context(0)
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234))
n = blocks.shape[0]
d_px = ctx.upload(blocks); d_out = ctx.alloc(n * 16)
for it in range(3):
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    uastc.encode_uastc_blocks(ctx, d_px, flags, n_blocks=n, out_device=d_out)
    ctx.sync()
    dt = time.perf_counter() - t0
    k = ctx.profile_read()
    print(f"run {it}: {dt*1e3:.1f} ms  {size*size/1e6/dt:.1f} Mpix/s ", {a: round(b[0], 2) for a, b in k.items()}, flush=True)
