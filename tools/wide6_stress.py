"""GPU-box helper: randomized trees through the 6-float many-workgroup path (BU_TSVQ_WIDE6_MIN=512) against the host restatement -- endpoint-like vectors of random
darkness, weight ranges from 1 to 2^40, random sizes and leaf budgets; every fourth tree has 150,000-240,000 vectors: up to the CEILING of what the endpoint builder can
ever be given (236,235 distinct vectors, tests/test_host_logic.py::test_endpoint_codebook_can_never_reach_the_threaded_gate). One line per mismatch and a summary; exit code 1 on any mismatch.
   usage: python tools/wide6_stress.py [seconds]"""
import os, sys, time, ctypes as C, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
os.environ["BU_TSVQ_WIDE6_MIN"] = "512"
import numpy as np
from basis_universal_amd import capi, etc1s
VP = C.c_void_p
ctx = capi.Context(0)
F = etc1s.load_frontend_library()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0


def expand(c5):
    return ((c5 << 3) | (c5 >> 2)).astype(np.float32) * np.float32(1.0 / 255.0)


t_end, cases, bad, seed, biggest = time.time() + budget, 0, 0, 0, 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(150000, 240001)) if seed % 4 == 0 else int(rng.integers(600, 90000))
    dark = rng.random()                      # share of near-black vectors
    span = int(rng.integers(1, 13)) if n < 100000 else int(rng.integers(6, 13))   # (enough distinct low / high pairs for the large trees)
    lo = rng.integers(0, 32, (n, 3)); hi = np.minimum(31, lo + rng.integers(0, span, (n, 3)))
    m = rng.random(n) < dark
    lo[m] = rng.integers(0, 3, (int(m.sum()), 3)); hi[m] = np.minimum(31, lo[m] + rng.integers(0, 3, (int(m.sum()), 3)))
    v = np.ascontiguousarray(np.unique(np.concatenate([expand(lo), expand(hi)], axis=1), axis=0))
    if rng.random() < 0.5:
        v = np.ascontiguousarray(v[rng.permutation(v.shape[0])])   # the builder takes the list as given: unsorted lists put dark vectors anywhere
    n = v.shape[0]
    wmax = int(2 ** rng.integers(0, 41))
    w = rng.integers(1, wmax + 1, n).astype(np.uint64)
    if rng.random() < 0.5:
        w[rng.random(n) < 0.8] = 1
    k = int(rng.integers(2, max(3, min(n, 3000)))); p = int(rng.integers(0, 33))
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32); a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32); st = np.zeros(3, np.uint32)
    assert F.bu_host_tsvq(6, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a1.ctypes.data_as(VP), cap, b1.ctypes.data_as(VP), cap) == 1
    ok = F.bu_device_tsvq(ctx.h, 6, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, k, p, a2.ctypes.data_as(VP), cap, b2.ctypes.data_as(VP), cap, st.ctypes.data_as(VP))
    cases += 1
    if ok != 1 or not ((a1 == a2).all() and (b1 == b2).all()):
        bad += 1
        print(f"MISMATCH seed {seed}: n {n} k {k} p {p} wmax 2^{int(np.log2(wmax))} dark {dark:.2f} ok {ok} leaves {a1[0]} vs {a2[0]}", flush=True)
    biggest = max(biggest, n)
print(f"{cases} trees (largest {biggest} vectors), {bad} mismatches")
sys.exit(1 if bad else 0)
