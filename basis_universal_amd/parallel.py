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


def rdo_strips(n_blocks, total_jobs):
    """The strips uastc_rdo cuts n_blocks into for total_jobs (encoder/basisu_uastc_enc.cpp:4103-4127): [(first, last), ...]."""
    per_job = n_blocks // total_jobs if total_jobs else 0
    if total_jobs <= 1 or per_job <= 8:
        return [(0, n_blocks)] if n_blocks else []
    return [(f, min(n_blocks, f + per_job)) for f in range(0, n_blocks, per_job)]


def uastc_rdo_sharded(rdo_fn, uastc_blocks, tiles, total_jobs, group=None):
    """uastc_rdo over the ranks: strips never look across their borders, so they are dealt out in contiguous runs (ceil(strips / world) per
    rank) and the modified blocks are gathered -- no exchange while the strips walk. rdo_fn(blocks, tiles, total_jobs) -> blocks is the
    single-process op (on a GPU rank: lambda b, t, j: uastc.uastc_rdo(ctx, b, t, params, flags, j)[0]). Every rank passes the whole arrays
    and gets the whole result, bit-identical to rdo_fn(uastc_blocks, tiles, total_jobs) on one process."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uastc_blocks = np.ascontiguousarray(uastc_blocks, np.uint8).reshape(-1, 16)
    tiles = np.ascontiguousarray(tiles, np.uint8).reshape(-1, 64)
    strips = rdo_strips(uastc_blocks.shape[0], total_jobs)
    per_rank = -(-len(strips) // world) if strips else 0
    mine = strips[rank * per_rank:(rank + 1) * per_rank]
    parts = []
    if mine:
        full = [s for s in mine if s[1] - s[0] == mine[0][1] - mine[0][0]]
        f0, f1 = full[0][0], full[-1][1]
        # k equal strips in one call reproduce themselves: (k * per_job) // k == per_job
        parts.append(rdo_fn(uastc_blocks[f0:f1], tiles[f0:f1], len(full) if len(full) > 1 else 0))
        for s in mine[len(full):]:  # the short last strip of the image, if it landed here
            parts.append(rdo_fn(uastc_blocks[s[0]:s[1]], tiles[s[0]:s[1]], 0))
    local = np.concatenate(parts) if parts else np.zeros((0, 16), np.uint8)
    sizes = [sum(b - a for a, b in strips[r * per_rank:(r + 1) * per_rank]) for r in range(world)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros((max(sizes + [1]), 16), dtype=torch.uint8, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return np.concatenate([out[r][:sizes[r]].cpu().numpy() for r in range(world)]) if strips else np.zeros((0, 16), np.uint8)
