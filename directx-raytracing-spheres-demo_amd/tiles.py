"""Tile partition of a frame over ranks -- the pure-Python statement of the layout the C-ABI uses
(pt_set_partition / pt_tiles_count / pt_render_tiles / pt_unpack_tiles), used by bench.py for buffer sizing and by the
tests as the reference for the device un-swizzle kernel.

The W x H frame is cut into ts x ts tiles in row-major tile order; tile t belongs to rank t % world and is that rank's
(t // world)-th tile.  A rank's packed buffer is [tiles_count(rank)][ts*ts] float4, row-major inside a tile; pixels of
edge tiles that fall outside the frame are zero.

Weighted partition (pt_set_partition_ex): a part is a residue range (first, run, stride) = the tiles t with
first <= t % stride < first + run, in increasing t; rank/world interleave = (rank, 1, world).  `weighted_partition`
gives the ranges for a root weight k: the assembling rank 0 owns k residues of every world - 1 + k, every other rank one."""
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


# ---- weighted partition (residue ranges) ------------------------------------------------------------------------------

def weighted_partition(rank, world, root_weight):
    """(first, run, stride) of `rank` when rank 0 carries `root_weight` shares and every other rank one.
    root_weight = 0: rank 0 renders the whole frame and the others nothing."""
    if world == 1 or root_weight == 0:
        return (0, 1, 1) if rank == 0 else (0, 0, 1)
    stride = world - 1 + root_weight
    return (0, root_weight, stride) if rank == 0 else (root_weight - 1 + rank, 1, stride)


def range_tile_ids(w, h, first, run, stride, ts=32):
    tx, ty = tile_grid(w, h, ts)
    return [t for t in range(tx * ty) if first <= t % stride < first + run]


def range_tiles_count(w, h, first, run, stride, ts=32):
    tx, ty = tile_grid(w, h, ts)
    total = tx * ty
    rem = total % stride
    return (total // stride) * run + (min(rem - first, run) if rem > first else 0)


def pack_range(frame, first, run, stride, ts=32, max_tiles=None):
    """frame (h, w, c) -> packed buffer (max_tiles or count, ts*ts, c) of the range's tiles, zero padded."""
    h, w, c = frame.shape
    tx, _ = tile_grid(w, h, ts)
    ids = range_tile_ids(w, h, first, run, stride, ts)
    out = np.zeros((max_tiles if max_tiles is not None else len(ids), ts * ts, c), dtype=frame.dtype)
    for k, t in enumerate(ids):
        x, y = t % tx, t // tx
        blk = np.zeros((ts, ts, c), dtype=frame.dtype)
        sub = frame[y * ts:(y + 1) * ts, x * ts:(x + 1) * ts]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        out[k] = blk.reshape(ts * ts, c)
    return out


def unpack_ranges(frame, parts, first0, run, stride, ts=32):
    """write parts[i] (tiles, ts*ts, c) = range (first0 + i*run, run, stride) into frame (h, w, c) in place: what
    pt_unpack_tiles_ex does on the device (pixels of other ranges stay untouched)."""
    h, w, _ = frame.shape
    tx, _ = tile_grid(w, h, ts)
    for i, part in enumerate(parts):
        for k, t in enumerate(range_tile_ids(w, h, first0 + i * run, run, stride, ts)):
            x, y = t % tx, t // tx
            blk = part[k].reshape(ts, ts, -1)
            hh, ww = min(ts, h - y * ts), min(ts, w - x * ts)
            frame[y * ts:y * ts + hh, x * ts:x * ts + ww] = blk[:hh, :ww]
    return frame
