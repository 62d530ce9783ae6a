"""Tile partition of a frame over ranks -- the pure-Python statement of the layout the C-ABI uses
(pt_set_partition / pt_tiles_count / pt_render_tiles / pt_unpack_tiles), used by bench.py for buffer sizing and by the
tests as the reference for the device un-swizzle kernel.

The W x H frame is cut into ts x ts tiles in row-major tile order; tile t belongs to rank t % world and is that rank's
(t // world)-th tile.  A rank's packed buffer is [tiles_count(rank)][ts*ts] float4, row-major inside a tile; pixels of
edge tiles that fall outside the frame are zero."""
import numpy as np


def tile_grid(w, h, ts=32):
    return (w + ts - 1) // ts, (h + ts - 1) // ts


def tiles_count(w, h, rank, world, ts=32):
    tx, ty = tile_grid(w, h, ts)
    total = tx * ty
    return (total - rank + world - 1) // world if rank < total else 0


def rank_tiles(w, h, rank, world, ts=32):
    """[(tile_x, tile_y)] of this rank's tiles in packed order."""
    tx, ty = tile_grid(w, h, ts)
    return [(t % tx, t // tx) for t in range(rank, tx * ty, world)]


def pack_tiles(frame, rank, world, ts=32, max_tiles=None):
    """frame (h, w, c) -> this rank's packed buffer (max_tiles or count, ts*ts, c), zero padded."""
    h, w, c = frame.shape
    tiles = rank_tiles(w, h, rank, world, ts)
    out = np.zeros((max_tiles if max_tiles is not None else len(tiles), ts * ts, c), dtype=frame.dtype)
    for k, (tx, ty) in enumerate(tiles):
        blk = np.zeros((ts, ts, c), dtype=frame.dtype)
        sub = frame[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        out[k] = blk.reshape(ts * ts, c)
    return out


def unpack_tiles(gathered, w, h, world, ts=32):
    """gathered (world, max_tiles, ts*ts, c) -> frame (h, w, c): what pt_unpack_tiles does on the device."""
    c = gathered.shape[-1]
    frame = np.zeros((h, w, c), dtype=gathered.dtype)
    for rank in range(world):
        for k, (tx, ty) in enumerate(rank_tiles(w, h, rank, world, ts)):
            blk = gathered[rank, k].reshape(ts, ts, c)
            hh, ww = min(ts, h - ty * ts), min(ts, w - tx * ts)
            frame[ty * ts:ty * ts + hh, tx * ts:tx * ts + ww] = blk[:hh, :ww]
    return frame
