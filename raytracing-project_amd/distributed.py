"""One frame shared by the GPUs of a node: every rank renders its own tiles,
only the framebuffer moves - one gather of the tile payloads to rank 0.

The ranks never exchange scene data (each holds the whole scene, it is
< 1 MB) and no collective sits inside the render: the gather at the end is the
path's only exchange step.  `torch.distributed` supplies the process group
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" for the CPU tests).
"""
import numpy as np

from . import tiles


def gather_frame(payload, W, H, tile_w, tile_h, rank, world, dist=None, unpack=None, deal=None):
    """Gathers every rank's (n_slots, 4) payload tensor on rank 0 and rebuilds the
    frame there.  `payload` is a torch tensor on the device the process group
    communicates on.  `unpack(src_rank, payload_tensor)` writes one rank's tiles
    into the caller's frame buffers (a GPU kernel on the GPU box, numpy in the
    CPU tests).  Ranks other than 0 return after the gather.  `deal` = (n_shares, [(first
    share, span) per rank]) for an uneven deal (tiles.shares); None = one share per rank."""
    import torch

    tile_px = tile_w * tile_h
    if deal is None:
        deal = (world, [(r, 1) for r in range(world)])
    max_slots = max(tiles.n_my_tiles(W, H, tile_w, tile_h, f, deal[0], s) for f, s in deal[1]) * tile_px
    if world == 1:
        unpack(0, payload)
        return
    if payload.shape[0] < max_slots:  # gather wants equal sizes: pad to the largest rank's share (deals may be uneven)
        pad = torch.zeros((max_slots - payload.shape[0], 4), dtype=payload.dtype, device=payload.device)
        payload = torch.cat([payload, pad])
    payload = payload.contiguous()
    if rank == 0:
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, gather_list=bufs, dst=0)
        for r in range(world):
            unpack(r, bufs[r])
    else:
        dist.gather(payload, gather_list=None, dst=0)


def render_cpu_partition(render_pixels, W, H, tile_w, tile_h, rank, world):
    """Helper for tests: payload of `rank` from a callable that shades a list of
    pixels, render_pixels(x, y) -> (rgb[n,3], depth[n])."""
    x, y, ok = tiles.slot_pixels(W, H, tile_w, tile_h, rank, world)
    out = np.zeros((len(x), 4), np.float32)
    if ok.any():
        rgb, depth = render_pixels(x[ok], y[ok])
        out[ok, :3] = rgb
        out[ok, 3] = depth
    return out
