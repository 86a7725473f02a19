"""Screen-space tile partition shared by the GPU kernels and the multi-GPU driver.

A frame is cut into tiles of tile_w x tile_h pixels (multiples of 8), numbered
row by row; tile t belongs to share t % world, and a rank owns `span` consecutive
shares starting at `rank` (span 1: tiles r, r + world, r + 2*world, ...).  A
rank's pixels are enumerated as "slots": tile after tile, and inside a tile by
8x8 blocks, so that the 64 lanes of a wavefront cover a compact screen area.
This module is the numpy statement of that enumeration (the device version is
slot_to_pixel in csrc/rt_device.h); the tile payload gathered to rank 0 is one
(r, g, b, depth) float4 per slot.
"""
import numpy as np


def n_tiles(W, H, tile_w, tile_h):
    return ((W + tile_w - 1) // tile_w) * ((H + tile_h - 1) // tile_h)


def n_my_tiles(W, H, tile_w, tile_h, rank, world, span=1):
    n = n_tiles(W, H, tile_w, tile_h)
    return (n // world) * span + max(0, min(n % world - rank, span))


def slot_pixels(W, H, tile_w, tile_h, rank, world, span=1):
    """(x, y, valid) for every slot of `rank`, in slot order."""
    tiles_x = (W + tile_w - 1) // tile_w
    tile_px = tile_w * tile_h
    n = n_my_tiles(W, H, tile_w, tile_h, rank, world, span) * tile_px
    slot = np.arange(n, dtype=np.int64)
    lt, within = slot // tile_px, slot % tile_px
    gt = (lt // span) * world + rank + lt % span
    tx, ty = gt % tiles_x, gt // tiles_x
    blk, inn = within >> 6, within & 63
    bpr = tile_w >> 3
    x = tx * tile_w + (blk % bpr) * 8 + (inn & 7)
    y = ty * tile_h + (blk // bpr) * 8 + (inn >> 3)
    return x, y, (x < W) & (y < H)


def pack(rgb, depth, tile_w, tile_h, rank, world, span=1):
    """Frame buffers -> (n_slots, 4) payload of `rank`."""
    H, W = depth.shape
    x, y, ok = slot_pixels(W, H, tile_w, tile_h, rank, world, span)
    out = np.zeros((len(x), 4), np.float32)
    out[ok, :3] = rgb[y[ok], x[ok]]
    out[ok, 3] = depth[y[ok], x[ok]]
    return out


def unpack(payload, rgb, depth, tile_w, tile_h, src_rank, world, span=1):
    """Payload of `src_rank` -> written into the frame buffers."""
    H, W = depth.shape
    x, y, ok = slot_pixels(W, H, tile_w, tile_h, src_rank, world, span)
    rgb[y[ok], x[ok]] = payload[: len(x)][ok, :3]
    depth[y[ok], x[ok]] = payload[: len(x)][ok, 3]


def shares(n_ranks, first_rank_weight=1.0, per_rank=8):
    """Uneven deal for n_ranks GPUs: every rank but the first owns `per_rank` shares, the first
    owns round(first_rank_weight * per_rank) (at least 1).  Returns (world, [(rank, span), ...])
    in the terms of RtParams: the i-th GPU renders with rank = first share, rank_span = span."""
    c0 = max(1, min(per_rank, int(round(first_rank_weight * per_rank))))
    if n_ranks == 1:
        return 1, [(0, 1)]
    spans = [c0] + [per_rank] * (n_ranks - 1)
    out, first = [], 0
    for s in spans:
        out.append((first, s))
        first += s
    return first, out
