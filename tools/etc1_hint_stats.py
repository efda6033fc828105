#!/usr/bin/env python3
"""How much of the ETC1 hint search (uastc_core.h etc1_fit_subblock) takes the integer form, on samples of the bench image and of the Kodak set. Host build, no GPU."""
import ctypes as C, pathlib, subprocess, sys
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers
so = pathlib.Path("/tmp/libuastc_host_stats.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DBU_ETC1_STATS", "-o", str(so), str(ROOT / "tests/native/uastc_host.cpp")])
L = C.CDLL(str(so))
u8p = C.POINTER(C.c_uint8)
L.hc_encode_uastc.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
def run(name, blocks, flags=2):
    blocks = np.ascontiguousarray(blocks)
    out = np.zeros((blocks.shape[0], 16), np.uint8)
    st = (C.c_ulonglong * 4)()
    L.hc_etc1_stats(st, 1)
    L.hc_encode_uastc(blocks.ctypes.data_as(u8p), blocks.shape[0], flags, out.ctypes.data_as(u8p))
    L.hc_etc1_stats(st, 1)
    t = list(st)
    print(f"{name:12s} blocks {blocks.shape[0]:6d}  tables: integer {t[0]:9d} general {t[1]:9d} ({t[0] / max(t[0] + t[1], 1):.3f})   errors: integer {t[2]:9d} general {t[3]:9d} ({t[2] / max(t[2] + t[3], 1):.3f})")
b = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
run("synth4096", b[:: b.shape[0] // 8192][:8192])
z = np.load(ROOT / "tests/golden/kodak24.npz")
for k in sorted(z.files)[:24:4]:
    img = np.concatenate([z[k], np.full(z[k].shape[:2] + (1,), 255, np.uint8)], axis=2)
    kb = helpers.to_pixel_blocks(img)
    run(k, kb[:: kb.shape[0] // 4096][:4096])
