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
