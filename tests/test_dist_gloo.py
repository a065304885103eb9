"""The N>1 path on CPU: tile partition + the single gather, world_size 2 and 3 over gloo.

No rendering here (there is no CPU render path); each rank fills its tile buffer with the
global pixel ids its tiles cover, the buffers are gathered exactly as bench.py does on
RCCL, and rank 0 de-interleaves with the same index map the HIP `untile` kernel implements.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ptamd
from ptamd.dist import TILE, gather_tiles, tile_counts, untile_index


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_tile_buffer(W, H, rank, world):
    """What pt_render_tiles would lay out for this rank, with pixel ids instead of radiance."""
    tx, ty, total, per_rank = tile_counts(W, H, world)
    buf = np.full((per_rank, TILE * TILE, 3), -1.0, np.float32)
    for lt in range(per_rank):
        tile = lt * world + rank
        if tile >= total:
            continue
        for lane in range(TILE * TILE):
            px, py = (tile % tx) * TILE + lane % TILE, (tile // tx) * TILE + lane // TILE
            if px < W and py < H:
                buf[lt, lane] = (py * W + px, rank, lt)
    return buf.reshape(-1)


def _worker(rank, world, port, W, H, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam = ptamd.make_camera(W, H)
        prm = ptamd.default_params(passes=1, spp_per_pass=1, rank=rank, world=world)
        n = ptamd.tiles_floats(cam, prm)                       # C-ABI host call, no GPU needed
        local = torch.from_numpy(_local_tile_buffer(W, H, rank, world))
        assert local.numel() == n
        gathered = gather_tiles(local, rank, world)
        if rank == 0:
            g = gathered.numpy().reshape(-1, 3)
            frame = g[untile_index(W, H, world)].reshape(H, W, 3)
            ok = np.array_equal(frame[..., 0], np.arange(W * H, dtype=np.float32).reshape(H, W))
            tx = (W + TILE - 1) // TILE
            ys, xs = np.divmod(np.arange(W * H), W)
            ok &= np.array_equal(frame[..., 1].ravel(), (((ys // TILE) * tx + xs // TILE) % world).astype(np.float32))
            q.put(bool(ok))
        else:
            assert gathered is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 64, 40), (3, 100, 52), (2, 13, 9)])
def test_gather_and_untile_over_gloo(world, W, H):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_tile_geometry_matches_c_abi():
    for W, H, world in ((1920, 1080, 8), (3840, 2160, 8), (13, 9, 2), (64, 64, 1), (100, 52, 3)):
        tx, ty, total, per_rank = tile_counts(W, H, world)
        for rank in range(world):
            prm = ptamd.default_params(passes=2, spp_per_pass=4, rank=rank, world=world)
            cam = ptamd.make_camera(W, H)
            assert ptamd.tiles_floats(cam, prm) == per_rank * 64 * 3
            assert ptamd.work_bytes(cam, prm) >= per_rank * 64 * 3 * 4 * 2      # staging slab + stream state of the pipeline
        idx = untile_index(W, H, world)
        assert len(np.unique(idx)) == W * H and idx.max() < world * per_rank * 64
    with pytest.raises(ptamd.PtError):
        ptamd.tiles_floats(ptamd.make_camera(16, 16), ptamd.default_params(rank=2, world=2))
