"""GPU-box helper: run the device TSVQ repeatedly against the host restatement, report mismatches and timings."""
import sys, time, ctypes as C, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from basis_universal_amd import capi, etc1s
import test_gpu_tsvq as T
VP = C.c_void_p
ctx = capi.Context(0)
F = etc1s.load_frontend_library()
def run(dim, n, k, p, kind, wmax, reps=3, host=True):
    rng = np.random.default_rng(n * 7 + k)
    v = T._data(kind, dim, n, rng); n = v.shape[0]
    w = rng.integers(1, wmax + 1, n).astype(np.uint64)
    if kind == "sel_skewed": w[rng.integers(0, n, 5)] = 3_000_000_000
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32)
    th = 0
    if host:
        t = time.time(); assert F.bu_host_tsvq(dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a1.ctypes.data_as(VP), cap, b1.ctypes.data_as(VP), cap) == 1; th = time.time() - t
    for mode in ("float", "packed"):
        if mode == "packed" and not kind.startswith("sel"): continue
        for r in range(reps):
            a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32)
            st = np.array([0xBACCED if mode == "packed" else 0, 0, 0], np.uint32)
            t = time.time()
            ok = F.bu_device_tsvq(ctx.h, dim, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a2.ctypes.data_as(VP), cap, b2.ctypes.data_as(VP), cap, st.ctypes.data_as(VP))
            td = time.time() - t
            same = bool((a1 == a2).all() and (b1 == b2).all()) if host else None
            msg = ""
            if host and not same:
                nl = int(a1[0]); o1 = a1[1:nl + 2]; o2 = a2[1:int(a2[0]) + 2]
                d = np.nonzero(o1[:min(len(o1), len(o2))] != o2[:min(len(o1), len(o2))])[0]
                msg = f" leaves {a1[0]} vs {a2[0]} first differing offset idx {d[:3]}"
            print(f"{kind} dim{dim} n{n} k{k} {mode} rep{r}: ok={ok} same={same} host {th:.3f}s device {td:.3f}s rounds {st[0]} splits {st[1]}/{st[2]}{msg}", flush=True)
for case in [(16, 30000, 900, 16, "sel_skewed", 4096), (16, 5000, 300, 32, "sel", 50), (16, 120000, 2731, 32, "sel", 4096)]:
    run(*case)
if len(sys.argv) > 1:
    run(16, 700000, 2731, 32, "sel", 4096, reps=2, host=False)
