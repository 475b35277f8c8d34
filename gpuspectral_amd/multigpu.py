"""Tile partition of a frame over ranks + the single gather of HDR tiles.

The scene and BVH are replicated on every GPU; pixels are independent (the RNG
seed depends only on the global pixel index and the timestamp, raygen.rgen:37),
so any partition reproduces the single-GPU image bit for bit.  Tiles of
TILE x TILE pixels are dealt round-robin for load balance.  The only collective
of a render is one gather of the ranks' compact RGBA32F buffers to rank 0
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import functools

import numpy as np

from . import pt

TILE = 32


@functools.lru_cache(maxsize=64)
def _ids(width, height, rank, world, tile):
    ids = pt.tile_partition(width, height, rank, world, tile)  # the C++ partition (gsp_tile_partition)
    ids.setflags(write=False)
    return ids


def tile_pixel_ids(width, height, rank, world, tile=TILE):
    """Pixels of the tiles owned by `rank`: the C ABI's gsp_tile_partition, cached and read-only."""
    return _ids(int(width), int(height), int(rank), int(world), int(tile))


def partition(width, height, rank, world, tile=TILE):
    """Sorted global pixel ids owned by `rank` (None when world == 1: whole frame)."""
    if world <= 1:
        return None
    return tile_pixel_ids(width, height, rank, world, tile)


def gather_frame(local_rgba, width, height, rank, world, dist, tile=TILE, force=False):
    """Gather the ranks' compact [n_r, 4] float32 tensors to rank 0 and assemble the
    full [height*width, 4] frame there (returns None on the other ranks).

    `local_rgba` is a torch tensor (cuda for nccl, cpu for gloo) holding this rank's
    pixels in partition() order.  force: a one-rank group still runs the collective (the RCCL
    smoke run of a one-GPU box, bench.py --force-dist)."""
    import torch

    if world <= 1 and not force:
        return local_rgba
    if world <= 1:
        gl = [torch.empty_like(local_rgba)]
        dist.gather(local_rgba, gl, dst=0)
        return gl[0]
    counts = [len(tile_pixel_ids(width, height, r, world, tile)) for r in range(world)]
    maxn = max(counts)
    buf = torch.zeros((maxn, 4), dtype=torch.float32, device=local_rgba.device)
    buf[: counts[rank]] = local_rgba[: counts[rank]]
    gl = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gl, dst=0)
    if rank != 0:
        return None
    frame = torch.zeros((height * width, 4), dtype=torch.float32, device=local_rgba.device)
    for r in range(world):
        frame[_index_tensor(width, height, r, world, tile, str(local_rgba.device))] = gl[r][: counts[r]]
    return frame


_INDEX_CACHE = {}


def _index_tensor(width, height, r, world, tile, device):
    """Pixel ids of rank r as an int64 tensor on `device` (built once: the gather sits in bench.py's timed region)."""
    import torch

    key = (width, height, r, world, tile, device)
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = torch.from_numpy(tile_pixel_ids(width, height, r, world, tile).astype(np.int64)).to(device)
    return _INDEX_CACHE[key]
