import sys, time, ctypes as C, pathlib, hashlib, os
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from basis_universal_amd import capi, etc1s
import test_gpu_tsvq as T
VP = C.c_void_p
ctx = capi.Context(0)
F = etc1s.load_frontend_library()
dim, n, k, p, kind, wmax = 16, 120000, 2731, 32, "sel", 4096
rng = np.random.default_rng(n * 7 + k)
v = T._data(kind, dim, n, rng); n = v.shape[0]
w = rng.integers(1, wmax + 1, n).astype(np.uint64)
cap = 4 * n + 4 * k + 100
a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32)
assert F.bu_host_tsvq(dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a1.ctypes.data_as(VP), cap, b1.ctypes.data_as(VP), cap) == 1
for mode in sys.argv[1:]:
    bad = 0; N = 12
    for r in range(N):
        a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32)
        st = np.array([0xBACCED if mode == "packed" else 0, 0, 0], np.uint32)
        F.bu_device_tsvq(ctx.h, dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a2.ctypes.data_as(VP), cap, b2.ctypes.data_as(VP), cap, st.ctypes.data_as(VP))
        if not ((a1 == a2).all() and (b1 == b2).all()): bad += 1
        # churn the allocator so the next run sees different recycled memory
        junk = [ctx.upload(np.full(100000 + 1000 * r, 0xA5, np.uint8)) for _ in range(3)]
        for j in junk: ctx.free(j)
    print(mode, "serial" if os.environ.get("BU_TSVQ_SERIAL") else "batched", "mismatches", bad, "of", N, flush=True)
