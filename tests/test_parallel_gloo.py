"""N > 1 path on CPU: world_size 2 (and 3, uneven slabs) over gloo. The kernels cannot run here, so the per-slab encoder is the host build
of the UASTC core (tests/native) -- what is under test is the slab partition and the result gather of basis_universal_amd/parallel.py."""
import os
import socket

import numpy as np
import pytest

import helpers
from basis_universal_amd import parallel


def test_slab_partition_covers_everything():
    for nby in (1, 2, 7, 8, 9, 1024):
        for world in (1, 2, 3, 8):
            rows = [parallel.slab_rows(nby, world, r) for r in range(world)]
            assert sum(n for _, n in rows) == nby
            pos = 0
            for first, n in rows:
                assert first == pos or n == 0
                pos += n
            assert max(n for _, n in rows) == -(-nby // world)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nbx, nby, tiles, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = parallel.encode_uastc_sharded(helpers.host_encode_uastc, tiles, nbx, nby, 2)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nbx,nby", [(2, 8, 6), (3, 5, 7)])
def test_sharded_uastc_equals_single_process(tmp_path, world, nbx, nby):
    import torch.multiprocessing as mp
    img = helpers.synth(nbx * 4, nby * 4, 321)
    tiles = helpers.to_pixel_blocks(img)
    want = helpers.host_encode_uastc(tiles, 2)
    mp.spawn(_worker, args=(world, _free_port(), nbx, nby, tiles, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == want.shape and (got == want).all(), f"rank {r}"


def test_rdo_strips_follow_the_reference_rule():
    assert parallel.rdo_strips(100, 0) == [(0, 100)] and parallel.rdo_strips(100, 1) == [(0, 100)]
    assert parallel.rdo_strips(100, 4) == [(0, 25), (25, 50), (50, 75), (75, 100)]
    assert parallel.rdo_strips(103, 4) == [(0, 25), (25, 50), (50, 75), (75, 100), (100, 103)]   # the loop steps by n // jobs (uastc_enc.cpp:4115)
    assert parallel.rdo_strips(40, 5) == [(0, 40)]  # blocks_per_job <= 8: one strip (uastc_enc.cpp:4109)
    assert parallel.rdo_strips(0, 4) == []


def _rdo_worker(rank, world, port, packed, tiles, jobs, lam, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = parallel.uastc_rdo_sharded(lambda b, t, j: helpers.host_uastc_rdo(b, t.reshape(-1, 4, 4, 4), 2, j, lam=lam), packed, tiles, jobs)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,jobs,n", [(2, 4, 600), (2, 4, 603), (3, 4, 603), (2, 0, 300), (3, 7, 500)])
def test_sharded_rdo_equals_single_process(tmp_path, world, jobs, n):
    """Strips dealt over 2 and 3 ranks (uneven, with and without the short last strip, single-strip mode): same bytes as one process."""
    import torch.multiprocessing as mp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "uastc_rdo_vectors.npz"))
    packed, tiles = g["packed_l2"][1500:1500 + n], g["blocks"][1500:1500 + n]
    want = helpers.host_uastc_rdo(packed, tiles, 2, jobs, lam=4.0)
    assert (want != packed).any()
    mp.spawn(_rdo_worker, args=(world, _free_port(), packed, tiles, jobs, 4.0, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == want.shape and (got == want).all(), f"rank {r}"
