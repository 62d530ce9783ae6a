"""Multi-GPU path without GPUs: the tile partition + gather + un-swizzle, exercised with world_size 2 and
3 over the gloo backend.  Each rank renders ITS tiles with the CPU oracle (test infrastructure standing in for the
HIP renderer), packs them exactly as pt_render_tiles lays them out, rank 0 gathers and un-swizzles; the result must be
bit-identical to the single-rank frame (the RNG is keyed on global pixel coordinates + frame)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_every_pixel_once(dxrs):
    from dxrs_amd import tiles
    for (w, h, world) in ((1920, 1080, 8), (3840, 2160, 8), (100, 70, 3), (31, 33, 2), (64, 64, 5)):
        seen = np.zeros((h, w), dtype=int)
        total = 0
        for rank in range(world):
            tl = tiles.rank_tiles(w, h, rank, world)
            assert len(tl) == tiles.tiles_count(w, h, rank, world)
            total += len(tl)
            for tx, ty in tl:
                seen[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] += 1
        assert (seen == 1).all() and total == np.prod(tiles.tile_grid(w, h))
        # interleaving balances the load: tile counts differ by at most one
        counts = [tiles.tiles_count(w, h, r, world) for r in range(world)]
        assert max(counts) - min(counts) <= 1
    frame = np.random.default_rng(0).random((70, 100, 4)).astype(np.float32)
    gathered = np.stack([tiles.pack_tiles(frame, r, 3, max_tiles=tiles.tiles_count(100, 70, 0, 3)) for r in range(3)])
    assert np.array_equal(tiles.unpack_tiles(gathered, 100, 70, 3), frame)


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import dxrs_amd_loader  # noqa: F401
    import dxrs_amd
    from dxrs_amd import tiles
    from oracle.binding import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host = dxrs_amd.load_host()
    oracle = load_oracle()
    spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_SMALL, seed=0)
    gs = dxrs_amd.types.graphics_settings(w, h, frame_index=3, bounces=4, spp=2)
    cam = host.camera(w, h, jitter_index=3)
    ts = 32
    max_tiles = tiles.tiles_count(w, h, 0, world)
    packed = np.zeros((max_tiles, ts * ts, 4), dtype=np.float32)
    for k, (tx, ty) in enumerate(tiles.rank_tiles(w, h, rank, world)):
        rw, rh = min(ts, w - tx * ts), min(ts, h - ty * ts)
        img, _ = oracle.render(spheres, materials, sd, cam, gs, rect=(tx * ts, ty * ts, rw, rh))
        blk = np.zeros((ts, ts, 4), dtype=np.float32)
        blk[:rh, :rw] = img
        packed[k] = blk.reshape(ts * ts, 4)
    t = torch.from_numpy(packed)
    gather_list = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gather_list, dst=0)
    if rank == 0:
        gathered = np.stack([g.numpy() for g in gather_list])
        frame = tiles.unpack_tiles(gathered, w, h, world)
        full, _ = oracle.render(spheres, materials, sd, cam, gs, threads=2)
        np.save(out_path, np.stack([frame, full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_unswizzle_bit_identical(world, tmp_path):
    import torch.multiprocessing as mp

    w, h = 100, 70  # ragged: edge tiles are partial, tile count not divisible by world
    out = str(tmp_path / "frames.npy")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, w, h, out), nprocs=world, join=True)
    frame, full = np.load(out)
    assert np.array_equal(frame.view(np.uint32), full.view(np.uint32))


def test_weighted_partition_covers_every_pixel_once(dxrs):
    """residue-range partition (pt_set_partition_ex): for every root weight the ranges of all ranks tile the frame exactly
    once, non-root shares are equal up to one tile, and pack_range / unpack_ranges round-trip a frame"""
    from dxrs_amd import tiles
    rng = np.random.default_rng(1)
    for (w, h, world) in ((1920, 1080, 8), (100, 70, 3), (31, 33, 2), (257, 129, 4)):
        tx, ty = tiles.tile_grid(w, h)
        for k in (0, 1, 2, 3, 5, 16):
            owner = np.full(tx * ty, -1)
            for rank in range(world):
                first, run, stride = tiles.weighted_partition(rank, world, k)
                ids = tiles.range_tile_ids(w, h, first, run, stride)
                assert len(ids) == tiles.range_tiles_count(w, h, first, run, stride)
                assert (owner[ids] == -1).all()
                owner[ids] = rank
            assert (owner >= 0).all()
            if k:
                others = [(owner == r).sum() for r in range(1, world)]
                assert max(others) - min(others) <= 1 and others[0] == max(others)
                # the root's share is k times a non-root share (up to rounding at the frame's end)
                assert abs((owner == 0).sum() - k * others[0]) <= k
            else:
                assert (owner == 0).all()
        frame = rng.random((h, w, 4)).astype(np.float32)
        k = 3
        out = np.zeros_like(frame)
        first, run, stride = tiles.weighted_partition(0, world, k)
        tiles.unpack_ranges(out, [tiles.pack_range(frame, first, run, stride)], first, run, stride)
        n_other = tiles.range_tiles_count(w, h, *tiles.weighted_partition(1, world, k))
        parts = [tiles.pack_range(frame, *tiles.weighted_partition(r, world, k), max_tiles=n_other) for r in range(1, world)]
        tiles.unpack_ranges(out, parts, k, 1, stride)
        assert np.array_equal(out, frame)


class _OracleOps:
    """ops protocol of dxrs_amd.exchange.TileExchange with the CPU oracle standing in for the HIP renderer"""

    def __init__(self, dxrs, oracle, host, w, h):
        self.dxrs, self.oracle, self.w, self.h = dxrs, oracle, w, h
        self.spheres, self.materials, self.sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
        self.cams = [host.camera(w, h, jitter_index=k) for k in range(8)]
        self.cache = {}

    def full(self, k):
        if k not in self.cache:
            gs = self.dxrs.types.graphics_settings(self.w, self.h, frame_index=k, bounces=3, spp=1)
            self.cache[k] = self.oracle.render(self.spheres, self.materials, self.sd, self.cams[k % 8], gs, threads=2)[0]
        return self.cache[k]

    def alloc(self, n_px, channels=4):
        import torch
        return torch.zeros((n_px, channels), dtype=torch.float32)

    def pack_rgb(self, src, n_px, dst):
        dst[:n_px] = src[:n_px, :3]

    def unpack_rgb(self, packed, offset_px, part_stride_px, n_parts, first0, run, stride, frame):
        import torch
        rgba = torch.cat([packed, torch.ones((packed.shape[0], 1), dtype=torch.float32)], 1)
        self.unpack(rgba, offset_px, part_stride_px, n_parts, first0, run, stride, frame)

    def set_range(self, first, run, stride):
        self.range = (first, run, stride)

    def render(self, k, out):
        import torch
        from dxrs_amd import tiles
        packed = tiles.pack_range(self.full(k), *self.range)
        out[: packed.shape[0] * 1024] = torch.from_numpy(packed.reshape(-1, 4))

    def unpack(self, packed, offset_px, part_stride_px, n_parts, first0, run, stride, frame):
        from dxrs_amd import tiles
        flat = packed.numpy()
        n = tiles.range_tiles_count(self.w, self.h, first0, run, stride)
        parts = [flat[offset_px + i * part_stride_px: offset_px + i * part_stride_px + n * 1024].reshape(n, 1024, 4) for i in range(n_parts)]
        tiles.unpack_ranges(frame.numpy().reshape(self.h, self.w, 4), parts, first0, run, stride)


class _OracleOpsPtGather(_OracleOps):
    """... with the exchange going through ops.gather_parts -- the branch that calls pt_gather on the GPU.  pt_gather's contract
    (include/pt_api.h) is emulated over gloo with raw pointers: every rank but the root sends nbytes; the root receives world - 1 parts,
    rank r's at recv + (r - 1) * nbytes, and contributes nothing."""

    def render_full(self, k, frame):
        """optional op: with it, root weight 0 on several ranks renders whole frames on rank 0 (no tiles, no un-swizzle)"""
        import torch
        frame[:] = torch.from_numpy(self.full(k).reshape(-1, 4))
        self.full_frames = getattr(self, "full_frames", 0) + 1

    def gather_parts(self, send, recv, nbytes):
        import ctypes

        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        mine = torch.zeros(nbytes, dtype=torch.uint8) if rank == 0 else send.contiguous().view(torch.uint8).reshape(-1)[:nbytes].clone()
        assert mine.numel() == nbytes
        parts = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        if rank == 0:
            for r in range(1, world):
                ctypes.memmove(recv.data_ptr() + (r - 1) * nbytes, parts[r].data_ptr(), nbytes)


def _exchange_worker(rank, world, port, w, h, out_path, pt_gather=False):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    import dxrs_amd_loader  # noqa: F401
    import dxrs_amd
    from dxrs_amd.exchange import TileExchange
    from oracle.binding import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = (_OracleOpsPtGather if pt_gather else _OracleOps)(dxrs_amd, load_oracle(), dxrs_amd.load_host(), w, h)
    batch = 3
    results = {}
    ex_plain = TileExchange(ops, w, h, rank, world, batch, rgb=False)  # float4 records on the wire
    ex = TileExchange(ops, w, h, rank, world, batch)                    # default: 12 bytes per pixel on the wire
    for k in range(batch):
        ex_plain.submit(k)
    if rank == 0:
        results["plain"] = np.stack([f.numpy().reshape(h, w, 4).copy() for f in ex_plain.frames])
    for weight in (1, 2, 5, 0):
        ex.configure(weight)
        got = []
        n_frames = 7  # two full batches + a partial one flushed by finish()
        for k in range(n_frames):
            ex.submit(k)
            if rank == 0 and (k + 1) % batch == 0:
                got += [f.numpy().reshape(h, w, 4).copy() for f in ex.frames]
        ex.finish()
        if rank == 0:
            got += [f.numpy().reshape(h, w, 4).copy() for f in ex.frames[: n_frames % batch]]
            results[weight] = np.stack(got)

    # autotune: every rank must end up with the same weight; here the "frame time" is made up so that weight 2 wins
    def run_frames(e):
        t0 = time.perf_counter()
        for k in range(batch):
            e.submit(k)
        e.finish()
        return {1: 3.0, 2: 1.0, 0: 2.0}[e.root_weight] + (time.perf_counter() - t0) * 1e-6 + 0.01 * rank
    log = {}
    chosen = ex.autotune(run_frames, dist.barrier, candidates=[1, 2, 0], log=log)
    assert chosen == 2 and ex.root_weight == 2
    if pt_gather:  # this ops class offers render_full: weight 0 took the direct path (7 frames + the autotune's 2 x 3), on rank 0 only
        assert getattr(ops, "full_frames", 0) == (7 + 6 if rank == 0 else 0), getattr(ops, "full_frames", 0)
    if rank == 0:
        ref = np.stack([ops.full(k) for k in range(7)])
        assert np.array_equal(results["plain"].view(np.uint32), ref[:3].view(np.uint32))
        np.save(out_path, np.stack([results[wt] for wt in (1, 2, 5, 0)] + [ref]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pt_gather", [(2, False), (3, False), (2, True), (3, True)])
def test_batched_weighted_exchange_bit_identical(world, pt_gather, tmp_path):
    """dxrs_amd.exchange.TileExchange (the code bench.py runs over RCCL) over gloo: batched gather, root-weighted
    partitions incl. 'root renders everything', partial final batch, and the autotune protocol -- through torch.distributed.gather
    and through the ops.gather_parts branch (pt_gather's contract: no root contribution, parts of ranks 1.. packed from recv on)"""
    import torch.multiprocessing as mp

    w, h = 100, 70
    out = str(tmp_path / "ex.npy")
    port = 31500 + (os.getpid() % 2000) + world + (10 if pt_gather else 0)
    mp.spawn(_exchange_worker, args=(world, port, w, h, out, pt_gather), nprocs=world, join=True)
    res = np.load(out)
    for i in range(4):
        assert np.array_equal(res[i].view(np.uint32), res[4].view(np.uint32)), f"weight case {i}"
