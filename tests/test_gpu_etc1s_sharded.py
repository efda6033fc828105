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
