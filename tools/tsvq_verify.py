"""GPU-box debugging aid: run the device TSVQ with BU_TSVQ_VERIFY=1 (every batched split is re-run serially and compared bit for
bit, tsvq_device.h::verify_batch) and against the host restatement. Usage: python tools/tsvq_verify.py [reps] [big]"""
import sys, os, time, ctypes as C, pathlib
os.environ["BU_TSVQ_VERIFY"] = "1"
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from basis_universal_amd import capi, etc1s
import test_gpu_tsvq as T
VP = C.c_void_p
ctx = capi.Context(0)
F = etc1s.load_frontend_library()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def case(dim, n, k, p, kind, wmax, reps, host=True, modes=("packed", "float")):
    rng = np.random.default_rng(n * 7 + k)
    v = T._data(kind, dim, n, rng); n = v.shape[0]
    w = rng.integers(1, wmax + 1, n).astype(np.uint64)
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32)
    if host:
        assert F.bu_host_tsvq(dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a1.ctypes.data_as(VP), cap, b1.ctypes.data_as(VP), cap) == 1
    first = None
    for mode in modes:
        for r in range(reps):
            a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32)
            st = np.array([0xBACCED if mode == "packed" else 0, 0, 0], np.uint32)
            t = time.time()
            ok = F.bu_device_tsvq(ctx.h, dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a2.ctypes.data_as(VP), cap, b2.ctypes.data_as(VP), cap, st.ctypes.data_as(VP))
            if first is None: first = (a2.copy(), b2.copy())
            same_host = bool((a1 == a2).all() and (b1 == b2).all()) if host else None
            same_first = bool((first[0] == a2).all() and (first[1] == b2).all())
            print(f"{kind} n{n} k{k} {mode} rep{r}: ok={ok} same_as_host={same_host} same_as_first={same_first} rounds {st[0]} splits {st[1]}/{st[2]} {time.time()-t:.2f}s", flush=True)
            junk = [ctx.upload(np.full(100000 + 1000 * r, 0xA5, np.uint8)) for _ in range(3)]
            for j in junk: ctx.free(j)


case(16, 120000, 2731, 32, "sel", 4096, reps)
if len(sys.argv) > 2:
    case(16, 700000, 2731, 32, "sel", 4096, 2, host=True, modes=("packed",))
