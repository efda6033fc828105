"""Time the UASTC RDO phases on the GPU (HIP events per phase): python tools/rdo_time.py [size] [total_jobs] [lambda] [cpu_sample_blocks]"""
import sys, time, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
import helpers
from basis_universal_amd import capi, uastc
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lam = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cpu_n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = capi.Context(0)
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234))
n = blocks.shape[0]
d_px = ctx.upload(blocks); d_enc = ctx.alloc(n * 16); d_out = ctx.alloc(n * 16)
uastc.encode_uastc_blocks(ctx, d_px, 2, n_blocks=n, out_device=d_enc)
p = uastc.RdoParams(m_lambda=lam)
for it in range(2):
    ctx.check(ctx.lib.memcpy_d2d(ctx.h, d_out, d_enc, n * 16) if hasattr(ctx.lib, "memcpy_d2d") else 1, "copy")
    if not hasattr(ctx.lib, "memcpy_d2d"):
        packed = ctx.download(d_enc, (n, 16), np.uint8)
        ctx.free(d_out); d_out = ctx.upload(packed)
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    _, info = uastc.uastc_rdo(ctx, d_out, d_px, p, 2, jobs, n_blocks=n)
    dt = time.perf_counter() - t0
    k = ctx.profile_read()
    print(f"run {it}: size {size} jobs {jobs} lambda {lam}: {dt*1e3:.1f} ms  {size*size/1e6/dt:.2f} Mpix/s  {dt/n*1e6:.2f} us/block", info,
          {a: round(b[0], 2) for a, b in k.items()}, flush=True)
if cpu_n and helpers.have_ref():
    packed = ctx.download(d_enc, (n, 16), np.uint8)[:cpu_n]
    t0 = time.perf_counter()
    want = helpers.ref_uastc_rdo(packed, blocks[:cpu_n], 2, 0, lam=lam)
    dt = time.perf_counter() - t0
    print(f"reference CPU, one strip of {cpu_n} blocks: {dt:.2f} s  {dt/cpu_n*1e6:.1f} us/block", flush=True)
    got, _ = uastc.uastc_rdo(ctx, packed, blocks[:cpu_n], p, 2, 0)
    print("match", bool((got == want).all()))
