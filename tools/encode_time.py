#!/usr/bin/env python3
"""Per-block ETC1S fit (bu_hip_k_encode_etc1s_blocks) on the bench image and on uniform noise, per (quality, metric): ms per 4096^2 launch.   usage: python tools/encode_time.py [size]"""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import helpers
from basis_universal_amd import capi
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
ctx.check(ctx.lib.set_stream(ctx.h, torch.cuda.current_stream().cuda_stream), "set_stream")
for name, img in (("synth", helpers.synth(size, size, 1234)), ("noise", helpers.uniform_random(size, size, 42))):
    blocks = helpers.to_pixel_blocks(img); n = blocks.shape[0]
    d = torch.from_numpy(blocks.reshape(n, 64)).to(dev)
    out = torch.empty((n, 8), dtype=torch.uint8, device=dev)
    for quality in (0, 1, 2, 3):
        for perceptual in (1, 0):
            for _ in range(2): ctx.check(ctx.lib.k_encode_etc1s_blocks(ctx.h, d.data_ptr(), n, quality, perceptual, out.data_ptr()), "k_encode")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): ctx.check(ctx.lib.k_encode_etc1s_blocks(ctx.h, d.data_ptr(), n, quality, perceptual, out.data_ptr()), "k_encode")
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"{name} {size}^2 quality {quality} perceptual {perceptual}: {dt * 1e3:.3f} ms")
