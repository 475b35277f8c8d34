"""N > 1 path on CPU: tile partition + the single gather (gpuspectral_amd/multigpu.py) over the
gloo backend, world_size 2 and 3.  The GPU renderer is replaced by the oracle here (this is a
test); what is under test is the partition / gather / assembly code bench.py runs on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import CORNELL_XML, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from gpuspectral_amd import multigpu
    from oracle import mitsuba_loader as ml
    from oracle import oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ml.load_scene(CORNELL_XML)
    ids = multigpu.partition(W, H, rank, world)
    local, _ = orc.Oracle(sc).render(W, H, spp=spp, pixel_ids=ids, threads=2)
    frame = multigpu.gather_frame(torch.from_numpy(local), W, H, rank, world, dist)
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_gather_reproduces_full_frame(tmp_path, oracle_mod, cornell, world):
    import torch.multiprocessing as mp

    W, H, spp = 80, 48, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, out), nprocs=world, join=True)
    frame = np.load(out)
    ref, _ = oracle_mod.Oracle(cornell).render(W, H, spp=spp)
    assert np.array_equal(frame, ref)


def test_partition_is_a_disjoint_cover():
    from gpuspectral_amd import multigpu

    for (W, H, world) in [(1920, 1080, 8), (100, 37, 3), (64, 64, 2), (33, 33, 4)]:
        parts = [multigpu.partition(W, H, r, world) for r in range(world)]
        allp = np.concatenate(parts)
        assert len(allp) == W * H and len(np.unique(allp)) == W * H
        for p in parts:
            assert (np.diff(p.astype(np.int64)) > 0).all()  # strictly increasing (gsp_frame_begin contract)
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 3 * 32 * 32  # balanced to a few tiles
    assert multigpu.partition(64, 64, 0, 1) is None


def test_cpp_partition_equals_numpy_statement():
    """gsp_tile_partition (C ABI, what every multi-GPU path uses) against the numpy statement of the same rule
    (scenes.tile_pixel_ids, used by the oracle-side tools): identical lists for every share."""
    from gpuspectral_amd import pt, scenes

    for (W, H, world, tile) in [(1920, 1080, 8, 32), (100, 37, 3, 32), (4096, 4096, 8, 32), (65, 31, 2, 16), (5, 5, 7, 32), (1, 1, 1, 32)]:
        for r in range(world):
            assert np.array_equal(pt.tile_partition(W, H, r, world, tile), scenes.tile_pixel_ids(W, H, r, world, tile)), (W, H, world, r)
    assert len(pt.tile_partition(64, 64, 5, 4)) == 0  # rank outside the world: empty
