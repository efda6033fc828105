"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol, fails loudly without a GPU, and the host-side
logic above the C ABI (TSVQ, quality mapping) matches the real reference where oracle/_ref is available."""
import ctypes as C
import pathlib

import numpy as np
import pytest

from helpers import have_ref, ref, ptr, u32p, u64p, f32p, synth, to_pixel_blocks


def test_library_exports_every_declared_symbol():
    from basis_universal_amd import capi
    lib = capi.load_library()
    declared = capi.declared_symbols()
    assert len(declared) >= 30
    for sym in declared:
        assert hasattr(lib.dll, sym), sym
    assert sorted(capi._SIGNATURES) == declared, "ctypes signature table out of sync with include/basisu_hip.h"


def test_frontend_library_loads():
    from basis_universal_amd import etc1s
    L = etc1s.load_frontend_library()
    for sym in ("bu_frontend_create", "bu_frontend_init", "bu_frontend_compress", "bu_frontend_get", "bu_etc1s_quality_to_clusters", "bu_host_tsvq"):
        assert hasattr(L, sym)


def test_frontend_library_exports_every_declared_symbol():
    """include/basisu_hip_frontend.h and include/basisu_hip_backend.h (both served by libbasisu_frontend.so)."""
    import pathlib, re
    from basis_universal_amd import etc1s
    L = etc1s.load_frontend_library()
    inc = pathlib.Path(__file__).resolve().parent.parent / "include"
    total = 0
    for header in ("basisu_hip_frontend.h", "basisu_hip_backend.h"):
        names = re.findall(r"BU_HIP_API\s+[^;(]*?\b(\w+)\s*\(", (inc / header).read_text())
        assert names, header
        for sym in names:
            assert hasattr(L, sym), (header, sym)
        total += len(names)
    assert total >= 20


def test_no_silent_cpu_fallback():
    """Without a GPU, creating a context must raise -- never fall back to a host implementation."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from basis_universal_amd import capi
    with pytest.raises(capi.HipError):
        capi.Context()
    lib = capi.load_library()
    assert lib.is_available() == 0
    assert not lib.create_context()
    # section-1/2 calls with a null context fail (return 0) instead of crashing
    assert lib.encode_etc1s_blocks(None, None, 1, 16) == 0
    assert lib.k_encode_etc1s_blocks(None, None, 0, 1, 1, None) == 0


def test_product_does_not_import_the_oracle():
    import pathlib, re
    root = pathlib.Path(__file__).resolve().parent.parent / "basis_universal_amd"
    for f in list(root.rglob("*.py")) + list(root.rglob("*.cpp")) + list(root.rglob("*.h")) + list(root.rglob("*.hip")):
        txt = f.read_text()
        assert not re.search(r"oracle[/_]|liboracle|ref_harness", txt), f"{f} references the oracle"


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("dim,n,k,p,kind", [(6, 3000, 256, 16, "ep"), (6, 20000, 1200, 16, "ep"), (16, 20000, 500, 32, "sel"),
                                            (16, 60000, 900, 16, "sel"), (16, 4000, 300, 0, "gauss"), (6, 10, 64, 16, "ep"), (16, 2, 8, 0, "sel"),
                                            (16, 300, 300, 32, "dup")])
def test_host_tsvq_matches_reference(dim, n, k, p, kind):
    from basis_universal_amd import etc1s
    F = etc1s.load_frontend_library()
    rng = np.random.default_rng(n + k)
    if kind == "sel":
        v = rng.integers(0, 4, (n, dim)).astype(np.float32)
    elif kind == "ep":
        v = rng.integers(0, 256, (n, dim)).astype(np.float32) * np.float32(1.0 / 255.0)
    elif kind == "dup":
        v = np.repeat(rng.integers(0, 4, (n // 3, dim)), 3, axis=0).astype(np.float32)  # heavy duplication -> few distinct rows
    else:
        v = rng.normal(0, 1, (n, dim)).astype(np.float32)
    v = np.ascontiguousarray(np.unique(v, axis=0))
    n = v.shape[0]
    w = rng.integers(1, 50, n).astype(np.uint64)
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32); a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32)
    assert ref().ref_tsvq(dim, ptr(v, f32p), ptr(w, u64p), n, k, p, 0, ptr(a1, u32p), cap, ptr(b1, u32p), cap) == 1
    assert F.bu_host_tsvq(dim, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, k, p, a2.ctypes.data_as(C.c_void_p), cap,
                          b2.ctypes.data_as(C.c_void_p), cap) == 1
    assert (a1 == a2).all() and (b1 == b2).all()


# The reference's multi-threaded configuration (generate_hierarchical_codebook_threaded_internal, enc.h:2086-2215): a T-leaf tree, then T independent
# trees of ceil(K / T) leaves and ceil(P / T) parents. Driven through the reference's _internal half so that small inputs take the partitioned path
# (the outer function only does above 262,144 distinct vectors, enc.h:2316 -- the last case goes through that gate with the real job pool).
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("dim,n,k,p,kind,threads", [(16, 20000, 500, 32, "sel", 8), (16, 20000, 500, 32, "sel", 2), (16, 6000, 512, 16, "sel", 4), (6, 20000, 1200, 16, "ep", 8),
                                                    (6, 3000, 256, 16, "ep", 3), (16, 4000, 300, 0, "gauss", 5), (16, 300, 300, 32, "sel", 8), (16, 300, 100, 32, "sel", 8),
                                                    (16, 1000, 2000, 32, "sel", 4), (6, 255, 64, 16, "ep", 4), (16, 5000, 256, 32, "sel", 16), (16, 5000, 256, 32, "sel", 24),
                                                    (6, 400, 64, 16, "line", 4)])
def test_host_tsvq_partitioned_matches_reference(dim, n, k, p, kind, threads):
    from basis_universal_amd import etc1s
    F = etc1s.load_frontend_library()
    rng = np.random.default_rng(n + k + threads)
    if kind == "sel":
        v = rng.integers(0, 4, (n, dim)).astype(np.float32)
    elif kind == "ep":
        v = rng.integers(0, 256, (n, dim)).astype(np.float32) * np.float32(1.0 / 255.0)
    elif kind == "line":   # degenerate covariance: the first tree may end with fewer leaves than threads (enc.h:2122)
        v = np.tile(rng.integers(0, 3, n).astype(np.float32)[:, None], (1, dim))
    else:
        v = rng.normal(0, 1, (n, dim)).astype(np.float32)
    v = np.ascontiguousarray(np.unique(v, axis=0))
    n = v.shape[0]
    w = rng.integers(1, 50, n).astype(np.uint64)
    cap = 4 * n + 4 * k + 100
    a1 = np.zeros(cap, np.uint32); b1 = np.zeros(cap, np.uint32); a2 = np.zeros(cap, np.uint32); b2 = np.zeros(cap, np.uint32)
    assert ref().ref_tsvq_mt(dim, ptr(v, f32p), ptr(w, u64p), n, k, p, threads, 1, ptr(a1, u32p), cap, ptr(b1, u32p), cap) == 1
    assert F.bu_host_tsvq_mt(dim, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, k, p, threads, 1, a2.ctypes.data_as(C.c_void_p), cap,
                             b2.ctypes.data_as(C.c_void_p), cap) == 1
    assert (a1 == a2).all() and (b1 == b2).all()
    if threads > 1 and n >= 256 and k >= threads * 16 and kind != "line":   # the partition really is a different codebook
        a0 = np.zeros(cap, np.uint32); b0 = np.zeros(cap, np.uint32)
        assert F.bu_host_tsvq(dim, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, k, p, a0.ctypes.data_as(C.c_void_p), cap, b0.ctypes.data_as(C.c_void_p), cap) == 1
        assert not ((a0 == a2).all() and (b0 == b2).all())


def test_endpoint_codebook_can_never_reach_the_threaded_gate():
    """The reference partitions a codebook build over its threads only from 262,144 DISTINCT training vectors up (enc.h:2316: `unique_vecs.size() < 65536*4 ? 1 : max_threads`).
    An endpoint training vector is the low / high colour of an ETC1S block (frontend.cpp:825-866, etc.h:543-565) -- a function of (colour5, intensity table) alone, 32^3 x 8 =
    262,144 combinations, and clamping at 0 / 255 makes many of them coincide: over ALL combinations the oracle's (reference-pinned) training vectors take 236,235 distinct
    values. So whatever the image, the endpoint builder is single-tree in both of the reference's thread configurations; only the selector builder (up to one distinct
    vector per block) ever takes the T-way path. 236,235 is also the largest node the 6-float many-workgroup passes can meet (tools/wide6_stress.py goes to 240,000)."""
    from helpers import oracle, ptr, f32p
    O = oracle()
    c = np.arange(32, dtype=np.uint8)
    r, g, b, t = np.meshgrid(c, c, c, np.arange(8, dtype=np.uint8), indexing="ij")
    blk = np.zeros((r.size, 8), np.uint8)
    blk[:, 0] = r.ravel() << 3; blk[:, 1] = g.ravel() << 3; blk[:, 2] = b.ravel() << 3          # base5, delta3 = 0
    blk[:, 3] = (t.ravel() << 5) | (t.ravel() << 2) | 2                                             # both intensity tables, differential bit
    v6 = np.zeros((blk.shape[0], 6), np.float32)
    O.orc_endpoint_training_vectors(ptr(blk), blk.shape[0], ptr(v6, f32p))
    distinct = np.unique(v6.view(np.uint32), axis=0).shape[0]
    assert blk.shape[0] == 262144 and distinct == 236235
    assert distinct < 65536 * 4      # = bu::kThreadedCodebookMinUnique (csrc/host/tsvq.h), the reference's gate


def test_reference_max_threads():
    """frontend.cpp:873-876 / 2195-2198"""
    from basis_universal_amd.etc1s import reference_max_threads as t
    assert t(False, 64, 64) == 0 and t(True, 64, 0) == 8 and t(True, 4, 0) == 4 and t(True, 64, 3) == 3 and t(True, 2, 16) == 2 and 1 <= t(True) <= 8


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
@pytest.mark.parametrize("w,h,quality", [(64, 64, 128), (256, 192, 128), (512, 512, 128), (256, 256, 255), (256, 256, 1), (128, 128, 64), (768, 512, 192)])
def test_quality_to_clusters_matches_reference(w, h, quality):
    """comp.cpp:3325-3379 through the real basis_compressor (which also runs the whole encode)."""
    from basis_universal_amd.etc1s import quality_to_clusters
    img = synth(w, h, 5)
    ep = np.zeros(1, np.uint32); sel = np.zeros(1, np.uint32); size = np.zeros(1, np.uint64)
    assert ref().ref_compress_etc1s(ptr(img), w, h, quality, 1, 1, ptr(ep, u32p), ptr(sel, u32p), None, 0, ptr(size, u64p)) == 1
    assert quality_to_clusters(quality, (w // 4) * (h // 4)) == (int(ep[0]), int(sel[0]))


def test_bench_contract_fields():
    import pathlib, re
    txt = (pathlib.Path(__file__).resolve().parent.parent / "bench.py").read_text()
    for field in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline", "cpu_baseline"):
        assert f'"{field}"' in txt, field


def test_rccl_communicator_library_exports_its_header():
    """include/basisu_hip_comm.h: every declared symbol is exported by libbasisu_rccl.so (loads without a GPU; no compute call here)"""
    import re
    import ctypes as C
    from basis_universal_amd import capi
    root = pathlib.Path(__file__).resolve().parent.parent
    names = sorted(set(re.findall(r"BU_HIP_API[^;(]*?\b(bu_rccl_\w+)\s*\(", (root / "include" / "basisu_hip_comm.h").read_text())))
    assert len(names) == 7
    capi.load_library()
    L = C.CDLL(str(root / "basis_universal_amd" / "lib" / "libbasisu_rccl.so"))
    for n in names:
        assert hasattr(L, n), n


def test_rccl_group_communicators_need_one_thread_each():
    """bu_rccl_comm_init_all's communicators live in one process; a host thread that walks the ranks of ONE collective in turn would wait for a part it has not issued
    yet. The library turns that into an error: a thread that has issued collective number e (or a later one) for rank j may not issue number e for rank i. Nothing is
    bound beyond that -- a rank's NEXT collective may come from any thread (executor pools), and a new thread is never mistaken for a finished one whose id it
    inherited. (No GPU needed: the group here has no RCCL communicator behind it, so a collective the rule lets through fails with "no communicator".)"""
    import ctypes as C
    import threading
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import _BuComm
    capi.load_library()
    root = pathlib.Path(__file__).resolve().parent.parent
    L = C.CDLL(str(root / "basis_universal_amd" / "lib" / "libbasisu_rccl.so"))
    L.bu_rccl_last_error.restype = C.c_char_p
    L.bu_rccl_comm_destroy.argtypes = [C.c_void_p]
    comms = (C.c_void_p * 3)()
    assert L.bu_rccl_debug_unconnected_group(3, comms) == 1
    views = []
    for c in comms:
        v = _BuComm()
        assert L.bu_rccl_comm_fill(C.c_void_p(c), C.byref(v)) == 1
        views.append(v)

    def gather(i):
        ok = views[i].all_gather(views[i].user, None, 16)
        return ok, L.bu_rccl_last_error().decode()

    ok, err = gather(0)                       # this thread issues collective #0 for rank 0: let through, then "no communicator"
    assert ok == 0 and "no communicator" in err
    ok, err = gather(1)                       # the same thread with #0 for rank 1: the rule
    assert ok == 0 and "already issued collective #0 for rank 0" in err and "thread of their own" in err
    ok, err = views[2].all_reduce_u64(views[2].user, None, 4), L.bu_rccl_last_error().decode()
    assert ok == 0 and "already issued collective #0 for rank 0" in err
    seen = {}
    for k in range(3):                        # short-lived threads one after the other (their ids get reused): #0 and #1 for rank 1, #0 for rank 2 -- all let through
        t = threading.Thread(target=lambda k=k: seen.__setitem__(k, gather(1) if k < 2 else gather(2)))
        t.start(); t.join()
        assert seen[k][0] == 0 and "no communicator" in seen[k][1], seen[k]
    t = threading.Thread(target=lambda: seen.update(r0=gather(0), r2=gather(2)))   # a pool thread takes rank 0's NEXT collective (#1) -- fine -- and then #1 of rank 2: the rule
    t.start(); t.join()
    assert seen["r0"][0] == 0 and "no communicator" in seen["r0"][1]
    assert seen["r2"][0] == 0 and "already issued collective #1 for rank 0" in seen["r2"][1]
    ok, err = gather(1)                       # the first thread again: rank 1 is at #2 by now, ahead of everything this thread has issued: let through
    assert ok == 0 and "no communicator" in err
    for c in comms:
        L.bu_rccl_comm_destroy(C.c_void_p(c))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not present")
def test_unified_quality_effort_matches_the_reference():
    """compress.unified_quality_effort = basis_compressor_params::set_format_mode_and_quality_effort (comp.cpp:76-205) for every quality 1..100 x effort 0..10
    and the 'not given' value -1, both LDR formats; the RDO lambda bit for bit (binary32)."""
    from helpers import ref_quality_effort
    from basis_universal_amd.compress import unified_quality_effort
    for q in [-1] + list(range(1, 101)):
        for e in range(-1, 11):
            rq, rl, _, _, _ = ref_quality_effort(False, q, e)
            got = unified_quality_effort(False, q, e)
            assert got.get("quality", -1) == rq and got["comp_level"] == rl, (q, e, got, rq, rl)
            _, _, flags, rdo, lam = ref_quality_effort(True, q, e)
            got = unified_quality_effort(True, q, e)
            assert got["uastc_level"] == flags and (got["uastc_rdo_lambda"] is not None) == rdo, (q, e, got, flags, rdo)
            if rdo:
                assert np.float32(got["uastc_rdo_lambda"]).tobytes() == np.float32(lam).tobytes(), (q, got["uastc_rdo_lambda"], lam)


def test_frontend_pipeline_scheduler_selftest():
    """bu_frontend_pipeline_* (include/basisu_hip_frontend.h): the cooperative-task scheduler without a GPU -- tasks keep a pattern on their own stacks across thousands
    of yields on the one driver thread, exceptions thrown and caught inside a task stay inside it, a task that ends in an exception is reported as a failed job."""
    import ctypes as C
    from basis_universal_amd import etc1s
    L = etc1s.load_frontend_library()
    L.bu_frontend_pipeline_selftest.argtypes = [C.c_uint32] * 4
    for lanes, tasks, yields, failing in [(1, 3, 10, 0), (4, 40, 1000, 3), (16, 200, 50, 10), (3, 7, 0, 1)]:
        assert L.bu_frontend_pipeline_selftest(lanes, tasks, yields, failing) == 1, (lanes, tasks, yields, failing)
    assert C.sizeof(etc1s._FrontendJob) == 48   # = sizeof(bu_frontend_job): two pointers + eight 32-bit fields
    assert L.bu_frontend_pipeline_create(0, 0) is None and L.bu_frontend_pipeline_create(0, 17) is None   # 1..16 lanes


def test_host_block_pool_recycles_large_blocks_and_leaves_the_process_alone():
    """csrc/host/block_pool.cpp: the library's own allocations of 1 MiB and more are recycled (second run of the same work maps nothing new), and nothing of it is
    visible outside the library: no operator new / delete among its dynamic symbols, no mallopt call left in its sources."""
    import pathlib, subprocess
    from basis_universal_amd import etc1s
    F = etc1s.load_frontend_library()
    F.bu_host_pool_stats.argtypes = [C.POINTER(C.c_uint64)]
    rng = np.random.default_rng(5)
    v = np.ascontiguousarray(np.unique(rng.integers(0, 4, (120000, 16)).astype(np.float32), axis=0))
    n = v.shape[0]
    w = np.ones(n, np.uint64)
    cap = 4 * n + 1000
    a = np.zeros(cap, np.uint32); b = np.zeros(cap, np.uint32)
    st = [(C.c_uint64 * 4)() for _ in range(3)]
    F.bu_host_pool_stats(st[0])
    for i in (1, 2):
        assert F.bu_host_tsvq(16, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, 8, 4, a.ctypes.data_as(C.c_void_p), cap, b.ctypes.data_as(C.c_void_p), cap) == 1
        F.bu_host_pool_stats(st[i])
    assert st[1][0] > st[0][0], "the first run maps its large blocks"
    assert st[2][0] == st[1][0] and st[2][1] > st[1][1], "the second run reuses them"
    assert st[2][2] <= st[2][3]
    # bu_host_pool_trim: what the pool keeps goes back to the kernel on request, and the next run maps afresh
    F.bu_host_pool_trim.restype = C.c_uint64
    held = st[2][2]
    assert held > 0 and F.bu_host_pool_trim() == held
    F.bu_host_pool_stats(st[0])
    assert st[0][2] == 0
    assert F.bu_host_tsvq(16, v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), n, 8, 4, a.ctypes.data_as(C.c_void_p), cap, b.ctypes.data_as(C.c_void_p), cap) == 1
    F.bu_host_pool_stats(st[1])
    assert st[1][0] > st[2][0], "after a trim the blocks are mapped again"
    syms = subprocess.run(["nm", "-D", "--defined-only", str(etc1s.FRONTEND_LIB_PATH)], stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
    assert all(" bu_" in s for s in syms if s.strip()), [s for s in syms if s.strip() and " bu_" not in s][:5]
    src = pathlib.Path(etc1s.__file__).parent / "csrc" / "host"
    assert not any("mallopt(" in p.read_text() for p in src.glob("*.cpp"))
