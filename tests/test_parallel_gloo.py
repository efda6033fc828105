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
