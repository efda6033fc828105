"""GPU-box debugging aid: create the device TSVQ object repeatedly on the same data and digest (a) the root record,
(b) the root split record, (c) the children's member lists, (d) a repeat of the root split on the same object."""
import sys, hashlib, ctypes as C, pathlib
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests"))
import numpy as np
from basis_universal_amd import capi
import test_gpu_tsvq as T
VP = C.c_void_p
ctx = capi.Context(0); L = ctx.lib
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(n0 * 7 + 2731)
v = T._data("sel", 16, n0, rng); n = v.shape[0]
w = rng.integers(1, 4097, n).astype(np.uint64)
keys = np.zeros(n, np.uint32)
for k in range(16): keys = (keys << np.uint32(2)) | v[:, k].astype(np.uint32)
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
for mode in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("packed", "float")):
    for r in range(reps):
        rootrec = np.zeros(20, np.uint32)
        if mode == "packed": q = L.tsvq_create_packed16(ctx.h, keys.ctypes.data_as(VP), w.ctypes.data_as(VP), n, rootrec.ctypes.data_as(VP))
        else: q = L.tsvq_create(ctx.h, 16, v.ctypes.data_as(VP), w.ctypes.data_as(VP), n, rootrec.ctypes.data_as(VP))
        assert q
        node = np.zeros(22, np.uint32)  # buf,start,count,pad, weight(2), origin[16]
        node[2] = n; node[4:6] = rootrec[16:18]; node[6:22] = rootrec[0:16]
        outs = []
        for rr in range(3):
            out = np.zeros(42, np.uint32)
            assert L.tsvq_split(ctx.h, q, node.ctypes.data_as(VP), 1, out.ctypes.data_as(VP)) == 1
            lists = np.zeros(n, np.uint32)
            assert L.tsvq_read_members(ctx.h, q, 1, 0, n, lists.ctypes.data_as(VP)) == 1
            out[3] = 0
            outs.append((h(out), int(out[1]), int(out[2]), h(lists)))
        print(mode, r, "root", h(rootrec[:19]), "var", rootrec[18:19].view(np.float32)[0], "splits", outs, flush=True)
        L.tsvq_destroy(ctx.h, q)
        junk = [ctx.upload(np.full(100000 + 1000 * r, 0xA5, np.uint8)) for _ in range(3)]
        for j in junk: ctx.free(j)
