"""Tile-split rendering across the GPUs of one node: one process per GPU, one gather.

The reference is single-device (no cudaSetDevice, no collectives: SURVEY.md F7).  Here the
frame is cut into 8x8 tiles numbered row-major; rank r renders tiles t with t % world == r
for all passes into a compact tile-major buffer (pt_render_tiles), and ONE collective — a
gather of the finished tile buffers to rank 0 over RCCL/xGMI — assembles the frame, which
rank 0 de-interleaves with pt_untile.  Pixels never communicate and a pixel's seed depends
only on (pixel offset, pass) (srcs/pathtracer.cu:71), so the N-GPU frame is bit-identical
to the 1-GPU frame.

torch is plumbing here: device buffers, the current HIP stream, torch.distributed.
"""
import numpy as np

from . import TILE, Scene, tiles_floats, untile, work_bytes


def tile_counts(W, H, world):
    tx, ty = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    total = tx * ty
    return tx, ty, total, (total + world - 1) // world


def untile_index(W, H, world):
    """Pure index math of pt_untile: for every pixel (row-major) the position of its float3
    in the rank-major concatenation of tile buffers.  Used by the CPU/gloo tests and as the
    specification of the HIP `untile` kernel."""
    tx, _, _, per_rank = tile_counts(W, H, world)
    py, px = np.divmod(np.arange(W * H, dtype=np.int64), W)
    tile = (py // TILE) * tx + (px // TILE)
    rank, lt = tile % world, tile // world
    lane = (py % TILE) * TILE + (px % TILE)
    return rank * (per_rank * TILE * TILE) + lt * (TILE * TILE) + lane


def gather_tiles(local_tiles, rank, world, group=None):
    """The single exchange step: gather every rank's tile buffer on rank 0.
    Returns the rank-major concatenation on rank 0, None elsewhere."""
    import torch.distributed as dist
    if world == 1:
        return local_tiles
    out = None
    if rank == 0:
        import torch
        out = torch.empty((world,) + tuple(local_tiles.shape), dtype=local_tiles.dtype, device=local_tiles.device)
        dist.gather(local_tiles, list(out.unbind(0)), dst=0, group=group)
        return out.reshape(-1)
    dist.gather(local_tiles, None, dst=0, group=group)
    return out


class TileRenderer:
    """Per-rank device buffers for a (camera, params) pair; reusable across steps."""

    def __init__(self, scene: Scene, cam, prm, device, work=None):
        """work: an existing scratch tensor to share (renders that run one after the other can use the same one); it must hold
        at least pt_work_bytes() of this (camera, params) pair."""
        import torch
        self.torch = torch
        self.scene, self.cam, self.prm = scene, cam, prm
        self.device = device
        self.n_floats = tiles_floats(cam, prm)
        self.tiles = torch.empty(self.n_floats, dtype=torch.float32, device=device)
        need = work_bytes(cam, prm) // 4
        if work is not None and work.numel() < need:
            raise ValueError("shared work buffer too small: %d < %d floats" % (work.numel(), need))
        self.work = work if work is not None else torch.empty(need, dtype=torch.float32, device=device)

    def render(self):
        """Enqueued on torch's current stream.  Blocks the calling thread until the render has drained in the default
        render path (pt_render_tiles polls the live-stream count, include/pt_api.h); asynchronous only in mode 0."""
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self.scene.render_tiles(self.cam, self.prm, self.tiles.data_ptr(), self.work.data_ptr(), stream)
        return self.tiles

    def assemble(self, gathered, world):
        """Rank 0: rank-major tile buffers -> (H, W, 3) frame on the device."""
        torch = self.torch
        frame = torch.empty((self.cam.H, self.cam.W, 3), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        untile(gathered.data_ptr(), self.cam, world, frame.data_ptr(), stream)
        return frame
