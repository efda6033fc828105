"""GPU-box helper: randomized selector-like trees (packed rows) through the many-workgroup path (tsvq_wide_min = 512; covariance through the maps or chained, pre-composed
windows on or off, chosen per tree) against the host restatement. Exit code 1 on any mismatch.   usage: python tools/wide16_stress.py [seconds]"""
import os, sys, time, ctypes as C, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
os.environ["BU_TSVQ_WIDE_MIN"] = "512"
import numpy as np
from basis_universal_amd import capi, etc1s
VP = C.c_void_p
ctx = capi.Context(0)
F = etc1s.load_frontend_library()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t_end, cases, bad, seed = time.time() + budget, 0, 0, 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(600, 150000))
    centres = rng.integers(0, 4, (int(rng.integers(4, 400)), 16))
    noise = rng.random() * 0.4
    v = np.clip(centres[rng.integers(0, centres.shape[0], n)] + (rng.random((n, 16)) < noise) * rng.integers(-1, 2, (n, 16)), 0, 3).astype(np.float32)
    v = np.ascontiguousarray(np.unique(v, axis=0))
    n = v.shape[0]
    wmax = int(2 ** rng.integers(0, 30))
    w = rng.integers(1, wmax + 1, n).astype(np.uint64)
    if rng.random() < 0.3:
        w[rng.integers(0, n, 5)] = 3_000_000_000
    k = int(rng.integers(2, max(3, min(n, 3000)))); p = int(rng.integers(0, 33))
    knobs = dict(tsvq_wide_min=512, tsvq_windows=1 if rng.random() < 0.5 else 2, tsvq_wide_cov_min=0 if rng.random() < 0.5 else 98304)
    ctx.set_tuning(**knobs)   # bu_hip_set_tuning: the paths this tree takes (all bit-identical)
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32); a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32); st = np.array([0xBACCED, 0, 0], np.uint32)
    assert F.bu_host_tsvq(16, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a1.ctypes.data_as(VP), cap, b1.ctypes.data_as(VP), cap) == 1
    ok = F.bu_device_tsvq(ctx.h, 16, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a2.ctypes.data_as(VP), cap, b2.ctypes.data_as(VP), cap, st.ctypes.data_as(VP))
    cases += 1
    if ok != 1 or not ((a1 == a2).all() and (b1 == b2).all()):
        bad += 1
        print(f"MISMATCH seed {seed}: n {n} k {k} p {p} wmax 2^{int(np.log2(wmax))} knobs {knobs} ok {ok} leaves {a1[0]} vs {a2[0]}", flush=True)
print(f"{cases} trees, {bad} mismatches")
sys.exit(1 if bad else 0)
