"""-m gpu: the slab / cluster sharded ETC1S frontend (bu_comm) on real kernels. The GPU box has one GPU, so two
ranks share cuda:0 and exchange through gloo; what is under test is the sharding itself: per-block stages by slab, per-cluster
stages by cluster share, all-gather / sum-merge of the results. Every rank must end with exactly the single-GPU state, which in turn
equals the reference's (golden digests)."""
import json
import os
import socket

import numpy as np
import pytest

import test_gpu_etc1s_frontend as T

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out_dir, wide_min=0):
    if wide_min:
        os.environ["BU_TSVQ_WIDE_MIN"] = str(wide_min)   # large-node TSVQ path inside the rank-distributed splits
    import torch
    import torch.distributed as dist
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import Etc1sFrontend, TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = capi.Context(0)
        comm = TorchComm()
        blocks, max_ep, max_sel, level, perceptual = T._params(case)
        fe = Etc1sFrontend(ctx, comm)
        fe.init(blocks, max_ep, max_sel, level, perceptual)
        fe.compress()
        digest = T._digest({k: fe.get(k) for k in T.STATE})
        with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
            json.dump({"digest": digest, "calls": comm.calls, "error": comm.error}, f)
        fe.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world,wide_min", [("synth256_l1", 2, 0), ("synth256_l3_flat", 2, 0), ("synth256_l4", 2, 0), ("synth512_q128", 3, 0), ("synth512_q128", 2, 512)])
def test_sharded_frontend_equals_single_gpu(tmp_path, case, world, wide_min):
    """per-block stages by slab, per-cluster stages by cluster share, and the TSVQ's node splits of every round shared out over the ranks (child
    lists + result records merged by one exact sum all-reduce)"""
    import torch.multiprocessing as mp
    golden = json.loads(T.GOLDEN.read_text())[case]["digests"]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path), wide_min), nprocs=world, join=True)
    for r in range(world):
        res = json.loads((tmp_path / f"r{r}.json").read_text())
        assert res["error"] == ""
        assert res["calls"]["all_gather"] > 0 and res["calls"]["all_reduce_u64"] > 0
        assert res["digest"] == golden, (r, {k: v[:10] for k, v in res["digest"].items() if v != golden[k]})


def _big_worker(rank, world, port, tiles_path, g, out_dir):
    import torch
    import torch.distributed as dist
    from basis_universal_amd import capi
    from basis_universal_amd.etc1s import Etc1sFrontend, TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BU_HOST_THREADS"] = str(max(2, min(8, (os.cpu_count() or 8) // world)))     # the node's host cores divided over the ranks, as bench.py does
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = capi.Context(0)
        comm = TorchComm()
        blocks = np.load(tiles_path, mmap_mode="r")
        fe = Etc1sFrontend(ctx, comm, max_threads=g.get("threads", 1))
        fe.init(np.ascontiguousarray(blocks), g["max_endpoint_clusters"], g["max_selector_clusters"], g["level"], g["perceptual"])
        fe.compress()
        digest = T._digest({k: fe.get(k) for k in T.STATE})
        with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
            json.dump({"digest": digest, "calls": comm.calls, "error": comm.error}, f)
        fe.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("synth4096_q128", 2), ("synth4096_q128", 8), ("synth4096_q128_t8", 8), ("synth8192_q255", 2), ("synth8192_q255_t8", 8)])
def test_baseline_configs_sharded_over_ranks(tmp_path, case, world):
    """BASELINE.json's configs[1] and configs[3] ("8192x8192 -q255, 8 x MI355X block-row shard + all-reduce for the global codebooks") in the form they are stated:
    ONE image sharded over 2 and over 8 ranks (here: processes sharing the box's one GPU, gloo in the place of RCCL), single-tree and with the reference's default
    8-thread codebook configuration (T = 8 independent trees whose rounds are shared out over the ranks). Every rank must end with the state the REFERENCE ends
    with (tests/golden/etc1s_big_digests.json: tools/gen_golden_big.py ran oracle/_ref)."""
    import pathlib
    import torch.multiprocessing as mp
    from helpers import synth, to_pixel_blocks
    g = json.loads((pathlib.Path(__file__).parent / "golden" / "etc1s_big_digests.json").read_text())[case]
    seed = {4096: 1234, 8192: 5678}[g["width"]]
    tiles = tmp_path / "tiles.npy"
    np.save(tiles, to_pixel_blocks(synth(g["width"], g["height"], seed)))
    mp.spawn(_big_worker, args=(world, _free_port(), str(tiles), g, str(tmp_path)), nprocs=world, join=True)
    tiles.unlink()
    for r in range(world):
        res = json.loads((tmp_path / f"r{r}.json").read_text())
        assert res["error"] == ""
        assert res["calls"]["all_gather"] > 0 and res["calls"]["all_reduce_u64"] > 0
        assert res["digest"] == g["frontend_digests"], (r, {k: v[:10] for k, v in res["digest"].items() if v != g["frontend_digests"][k]})


def _rccl_worker(rank, out_dir):
    import ctypes as C
    from basis_universal_amd import capi, etc1s
    ctx = capi.Context(0)
    L = etc1s.load_rccl_library()
    ident = C.create_string_buffer(128)
    assert L.bu_rccl_get_unique_id(ident) == 1, L.bu_rccl_last_error()
    h = L.bu_rccl_comm_create(ctx.h, ident.raw, 0, 1)
    assert h, L.bu_rccl_last_error()
    comm = etc1s._BuComm()
    assert L.bu_rccl_comm_fill(h, C.byref(comm)) == 1 and comm.world == 1 and comm.rank == 0
    data = np.arange(4096, dtype=np.uint64) * np.uint64(0x0101010101010101)
    d = ctx.upload(data)
    assert comm.all_gather(comm.user, d, data.nbytes) == 1, L.bu_rccl_last_error()
    assert comm.all_reduce_u64(comm.user, d, data.size) == 1, L.bu_rccl_last_error()
    back = ctx.download(d, data.shape, np.uint64)
    assert (back == data).all()
    ctx.free(d)
    L.bu_rccl_comm_destroy(h)
    ctx.close()
    with open(os.path.join(out_dir, "rccl_ok"), "w") as f:
        f.write("ok")


def test_native_rccl_communicator_single_rank(tmp_path):
    """include/basisu_hip_comm.h on the box's one GPU: a real RCCL communicator (world 1) behind bu_comm; the two collectives run on the
    context's stream from C++ and leave a one-rank buffer as it is. (More ranks need more GPUs: RCCL refuses two ranks on one device.) In a
    process of its own, like every rank of a real job: RCCL keeps helper threads and its own view of the HIP runtime alive in whoever creates it."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(str(tmp_path),), nprocs=1, join=True)
    assert (tmp_path / "rccl_ok").read_text() == "ok"
