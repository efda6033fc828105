"""Multi-GPU sharding of the per-block stages: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests). torch is used for the process group only; tiles and results cross the C ABI as plain pointers.

Partition: contiguous block-row slabs (block index = by * num_blocks_x + bx, comp.cpp:3261), ceil(num_blocks_y / world) rows per rank,
the last ranks may get fewer (or zero) rows. Per-block ops (UASTC encode, ETC1S block encode / refine / selector assignment) need no
exchange while they run; their fixed-size per-block results are gathered with one all_gather.
"""
import numpy as np


def slab_rows(num_blocks_y, world, rank):
    """(first_row, n_rows) of this rank's slab."""
    per = -(-num_blocks_y // world)
    first = min(rank * per, num_blocks_y)
    return first, min(per, num_blocks_y - first)


def slab_blocks(num_blocks_x, num_blocks_y, world, rank):
    """(first_block, n_blocks) in block-raster order."""
    first, rows = slab_rows(num_blocks_y, world, rank)
    return first * num_blocks_x, rows * num_blocks_x


def gather_block_results(local, num_blocks_x, num_blocks_y, group=None):
    """all_gather of per-block results (n_local, k) uint8 arrays whose sizes follow slab_blocks(); returns the (N, k) array on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    k = local.shape[1]
    per = -(-num_blocks_y // world) * num_blocks_x  # padded slab size: all_gather wants equal shapes
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros((per, k), dtype=torch.uint8, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = []
    for r in range(world):
        _, n = slab_blocks(num_blocks_x, num_blocks_y, world, r)
        parts.append(out[r][:n].cpu().numpy())
    return np.concatenate(parts) if parts else np.zeros((0, k), np.uint8)


def encode_uastc_sharded(encode_fn, tiles, num_blocks_x, num_blocks_y, flags, group=None):
    """Every rank encodes its slab of `tiles` ((N, 4, 4, 4) uint8, block-raster order) with encode_fn(slab_tiles, flags) -> (n, 16) uint8
    (on a GPU rank: lambda t, f: uastc.encode_uastc_blocks(ctx, t, f)) and gathers the whole image's blocks."""
    import torch.distributed as dist
    first, n = slab_blocks(num_blocks_x, num_blocks_y, dist.get_world_size(group), dist.get_rank(group))
    local = encode_fn(tiles[first:first + n], flags) if n else np.zeros((0, 16), np.uint8)
    return gather_block_results(local, num_blocks_x, num_blocks_y, group)
