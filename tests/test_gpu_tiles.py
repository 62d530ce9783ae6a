"""Partition invariance: a frame rendered as interleaved 32x32 tiles by 1, 2, 3 or 8 ranks and
un-swizzled by pt_unpack_tiles is BIT-IDENTICAL to the single full-frame render.  The ranks are emulated one after
another on the one GPU of the test box (the gather is a torch.cat standing in for the RCCL gather)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def tiled_render(dxrs, r, w, h, world, torch):
    from dxrs_amd import tiles
    ts = 32
    max_tiles = tiles.tiles_count(w, h, 0, world)
    gathered = torch.zeros((world, max_tiles * ts * ts, 4), dtype=torch.float32, device="cuda")
    rays = 0
    for rank in range(world):
        r.set_partition(rank, world)
        assert r.tiles_count(rank) == tiles.tiles_count(w, h, rank, world)
        st = r.render_tiles(gathered[rank].data_ptr(), want_stats=True)
        rays += st.rays
    frame = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    r.set_partition(0, world)
    r.unpack_tiles(gathered.data_ptr(), max_tiles, frame.data_ptr())
    r.synchronize()
    # device un-swizzle == the Python statement of the layout
    ref = tiles.unpack_tiles(gathered.cpu().numpy().reshape(world, max_tiles, ts * ts, 4), w, h, world)
    out = frame.cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    r.set_partition(0, 1)
    return out, rays


@pytest.mark.parametrize("w,h,spp,bounces,worlds", [(1920, 1080, 1, 8, (1, 2, 8)), (200, 150, 3, 4, (1, 2, 3, 8)), (3840, 2160, 2, 8, (8,)),
                                                        (3840, 2160, 64, 16, (8,))])  # the last one is BASELINE config C4 as its 8 ranks render it
def test_partition_invariance(dxrs, host, renderer, w, h, spp, bounces, worlds):
    import torch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    gs = dxrs.types.graphics_settings(w, h, frame_index=11, bounces=bounces, spp=spp)
    cam = host.camera(w, h, jitter_index=11)
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(cam); renderer.set_constants(gs)
    full, st = renderer.render()
    for world in worlds:
        img, rays = tiled_render(dxrs, renderer, w, h, world, torch)
        assert rays == st.rays
        assert np.array_equal(img.view(np.uint32)[..., :3], full.view(np.uint32)[..., :3]), f"world={world}"


@pytest.mark.parametrize("w,h,world,weight", [(1920, 1080, 8, 3), (1920, 1080, 2, 4), (200, 150, 3, 2), (200, 150, 4, 1), (333, 97, 5, 7), (200, 150, 3, 0)])
def test_weighted_partition_invariance(dxrs, host, renderer, w, h, world, weight):
    """root-weighted residue-range partition (pt_set_partition_ex / pt_unpack_tiles_ex): ranks emulated one after another;
    the frame assembled from the root's own range + the gathered ranges is bit-identical to the full-frame render, and the
    device un-swizzle agrees with the Python statement of the layout"""
    import torch
    from dxrs_amd import tiles
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    gs = dxrs.types.graphics_settings(w, h, frame_index=5, bounces=6, spp=1)
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(host.camera(w, h, jitter_index=5)); renderer.set_constants(gs)
    renderer.set_partition(0, 1)
    full, st = renderer.render()
    ts2 = 32 * 32
    root = tiles.weighted_partition(0, world, weight)
    n_root = tiles.range_tiles_count(w, h, *root)
    n_other = tiles.range_tiles_count(w, h, *tiles.weighted_partition(1, world, weight)) if weight else 0
    assert renderer.tiles_count_ex(*root) == n_root
    own = torch.zeros((n_root * ts2, 4), dtype=torch.float32, device="cuda")
    others = torch.zeros((max(world - 1, 1), max(n_other, 1) * ts2, 4), dtype=torch.float32, device="cuda")
    rays = 0
    renderer.set_partition_ex(*root)
    rays += renderer.render_tiles(own.data_ptr(), want_stats=True).rays
    for rank in range(1, world):
        rng_ = tiles.weighted_partition(rank, world, weight)
        assert renderer.tiles_count_ex(*rng_) == tiles.range_tiles_count(w, h, *rng_)
        renderer.set_partition_ex(*rng_)
        stt = renderer.render_tiles(others[rank - 1].data_ptr(), want_stats=True)
        rays += stt.rays
    frame = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
    renderer.unpack_tiles_ex(own.data_ptr(), 0, 1, root[0], root[1], root[2], frame.data_ptr())
    if weight:
        renderer.unpack_tiles_ex(others.data_ptr(), others.shape[1], world - 1, root[1], 1, root[2], frame.data_ptr())
    renderer.synchronize()
    out = frame.cpu().numpy()
    assert rays == st.rays
    assert np.array_equal(out.view(np.uint32)[..., :3], full.view(np.uint32)[..., :3])
    ref = np.full((h, w, 4), -1.0, dtype=np.float32)
    tiles.unpack_ranges(ref, [own.cpu().numpy().reshape(n_root, ts2, 4)], *root)
    if weight:
        tiles.unpack_ranges(ref, list(others.cpu().numpy().reshape(world - 1, -1, ts2, 4)), root[1], 1, root[2])
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    if weight:
        # the 12-byte exchange format: pack the gathered parts to 3 floats per pixel, un-swizzle with alpha = 1 -> same frame
        rgb = torch.zeros((others.shape[0], others.shape[1], 3), dtype=torch.float32, device="cuda")
        renderer.pack_rgb(others.data_ptr(), others.shape[0] * others.shape[1], rgb.data_ptr())
        frame2 = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        renderer.unpack_tiles_ex(own.data_ptr(), 0, 1, root[0], root[1], root[2], frame2.data_ptr())
        renderer.unpack_tiles_rgb(rgb.data_ptr(), others.shape[1], world - 1, root[1], 1, root[2], frame2.data_ptr())
        renderer.synchronize()
        assert np.array_equal(rgb.cpu().numpy(), others.cpu().numpy()[..., :3])
        assert np.array_equal(frame2.cpu().numpy().view(np.uint32), out.view(np.uint32))
    with pytest.raises(RuntimeError):
        renderer.set_partition_ex(3, 2, 4)  # first + run > stride
    renderer.set_partition(0, 1)


def test_full_frame_determinism_and_counts(dxrs, host, renderer):
    """Full BASELINE C2 frame: run-to-run bit-identical although the queue order is decided by atomics; the ray count is
    consistent with the structure of the estimator (primaries + at most spp*bounces per pixel)."""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 1920, 1080
    gs = dxrs.types.graphics_settings(w, h, frame_index=2, bounces=8, spp=1)
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(host.camera(w, h, jitter_index=2)); renderer.set_constants(gs)
    a, sa = renderer.render()
    b, sb = renderer.render()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.rays == sb.rays
    assert sa.pixels == w * h and sa.paths == w * h and w * h < sa.rays <= w * h * 9
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 1).all()
    # the sky rows at the top of the frame are pure environment: smooth and bluish
    assert a[:8, :, 2].min() > a[:8, :, 0].max()


@pytest.mark.parametrize("lanes", [2, 3, 5])
def test_frames_in_flight(dxrs, host, renderer, lanes):
    """Frames in flight: consecutive frames run on N internal streams with N sets of work buffers and overlap on the
    GPU; with the caller rotating over N output buffers every frame is bit-identical to the one-frame-at-a-time render,
    and the device-accumulated ray totals agree."""
    import torch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h, n_frames = 640, 360, 12
    cams = [host.camera(w, h, jitter_index=k) for k in range(8)]
    gs = dxrs.types.graphics_settings(w, h, bounces=8, spp=1)
    renderer.set_scene(spheres, materials, sd)
    ref, ref_rays = [], 0
    for k in range(n_frames):
        gs.FrameIndex = k
        renderer.set_camera(cams[k % 8]); renderer.set_constants(gs)
        img, st = renderer.render()
        ref.append(img); ref_rays += st.rays
    tstream = torch.cuda.Stream()  # the caller's stream: consumers queued on it are ordered after each frame
    r2 = dxrs.Renderer(stream=tstream.cuda_stream, flags=dxrs.types.PT_FLAG_TWO_FRAMES_IN_FLIGHT) if lanes == 2 else \
        dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=lanes)
    try:
        with torch.cuda.stream(tstream):
            r2.set_scene(spheres, materials, sd)
            bufs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
            keep = []
            r2.totals(reset=True)
            for k in range(n_frames):
                gs.FrameIndex = k
                r2.set_camera(cams[k % 8]); r2.set_constants(gs)
                r2.render_device(bufs[k % lanes].data_ptr())
                keep.append(bufs[k % lanes].clone())  # a consumer on the caller's stream: ordered after frame k, before frame k + lanes
            torch.cuda.synchronize()
            tot = r2.totals()
            assert tot.rays == ref_rays
            for k in range(n_frames):
                assert np.array_equal(keep[k].cpu().numpy().view(np.uint32), ref[k].view(np.uint32)), f"frame {k}"
            # spp > 1 (host-polled passes) also works in this mode
            gs2 = dxrs.types.graphics_settings(w, h, frame_index=3, bounces=4, spp=3)
            renderer.set_camera(cams[3]); renderer.set_constants(gs2)
            a, _ = renderer.render()
            r2.set_camera(cams[3]); r2.set_constants(gs2)
            b, _ = r2.render()
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        r2.close()


def test_everything_together_tiles_textures_di_in_flight(dxrs, host, oracle):
    """the optional pieces composed: a weighted tile partition rendered with three frames in flight, textured spheres with
    per-frame rotations, sphere-light direct illumination, spp 2 -- every assembled frame equals the oracle's full frame"""
    import torch
    from dxrs_amd import tiles
    t = dxrs.types
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h, world, weight, n_frames = 320, 200, 3, 2, 5
    tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream)
    root = tiles.weighted_partition(0, world, weight)
    n_root = tiles.range_tiles_count(w, h, *root)
    n_other = tiles.range_tiles_count(w, h, *tiles.weighted_partition(1, world, weight))
    ts2 = 32 * 32
    rs = [dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=3) for _ in range(world)]  # one context per emulated rank
    for rank, r in enumerate(rs):
        r.set_scene(spheres, materials, sd)
        r.set_textures(host.demo_textures(0, 0.0))
        r.set_partition_ex(*tiles.weighted_partition(rank, world, weight))
    own = [torch.zeros((n_root * ts2, 4), dtype=torch.float32, device="cuda") for _ in range(n_frames)]
    others = [torch.zeros((world - 1, n_other * ts2, 4), dtype=torch.float32, device="cuda") for _ in range(n_frames)]
    frames = [torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda") for _ in range(n_frames)]
    tex = [host.demo_textures(0, 0.5 * k) for k in range(n_frames)]
    gss = [t.graphics_settings(w, h, frame_index=k, bounces=4, spp=2, di=True) for k in range(n_frames)]
    cams = [host.camera(w, h, jitter_index=k) for k in range(n_frames)]
    torch.cuda.synchronize()  # the fills above are done before any frame (a frame on another lane does not wait for work queued just before it)
    for k in range(n_frames):
        for rank, r in enumerate(rs):
            r.update_rotations(tex[k].rotations)
            r.set_camera(cams[k]); r.set_constants(gss[k])
            r.render_tiles((own[k] if rank == 0 else others[k][rank - 1]).data_ptr())
        rs[0].unpack_tiles_ex(own[k].data_ptr(), 0, 1, root[0], root[1], root[2], frames[k].data_ptr())
        rs[0].unpack_tiles_ex(others[k].data_ptr(), others[k].shape[1], world - 1, root[1], 1, root[2], frames[k].data_ptr())
    for r in rs:
        r.synchronize()
    torch.cuda.synchronize()
    for k in range(n_frames):
        ref, _ = oracle.render(spheres, materials, sd, cams[k], gss[k], threads=8, textures=tex[k])
        got = frames[k].cpu().numpy()
        assert np.array_equal(got.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"frame {k}"
    for r in rs:
        r.close()


@pytest.mark.gpu
def test_unsharded_candidate_renders_whole_frames(dxrs, host, renderer):
    """Root weight 0 with several ranks ("do not shard", one of the partitions bench.py's autotune tries): rank 0 renders whole frames
    straight into the exchange's frame buffers -- no tiles, no collective, no un-swizzle -- so rank 0 of a 2-rank job can be exercised
    here without a second rank; switching back to a sharded weight afterwards must find the buffers and the partition intact."""
    import torch
    from dxrs_amd.exchange import HipOps, TileExchange
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h, batch = 320, 200, 3
    cams = [host.camera(w, h, jitter_index=k) for k in range(8)]
    gs = dxrs.types.graphics_settings(w, h, bounces=6, spp=1)
    renderer.set_scene(spheres, materials, sd)
    ref = []
    for k in range(7):
        gs.FrameIndex = k
        renderer.set_camera(cams[k % 8]); renderer.set_constants(gs)
        ref.append(renderer.render()[0])
    tstream = torch.cuda.Stream()
    r2 = dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=3)
    try:
        with torch.cuda.stream(tstream):
            r2.set_scene(spheres, materials, sd)

            def set_frame(k):
                gs.FrameIndex = k
                r2.set_camera(cams[k % 8]); r2.set_constants(gs)
            ex = TileExchange(HipOps(r2, torch.device("cuda", 0), set_frame), w, h, 0, 2, batch)
            ex.configure(0)
            assert ex.direct and not ex.sharded
            got = []
            for k in range(7):
                ex.submit(k)
                if (k + 1) % batch == 0:
                    torch.cuda.synchronize()
                    got += [f.cpu().numpy().reshape(h, w, 4).copy() for f in ex.frames]
            ex.finish()
            torch.cuda.synchronize()
            got += [f.cpu().numpy().reshape(h, w, 4).copy() for f in ex.frames[: 7 % batch]]
            for k in range(7):
                assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"frame {k}"
            # back to a sharded partition: rank 0's share of the tiles renders as before (the collective itself needs the other rank)
            ex.configure(2)
            assert ex.sharded and not ex.direct and ex.own_px == ex.n_root * 1024
    finally:
        r2.close()


def test_bench_control_flow_with_two_ranks():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), rehearsed on ONE GPU: RCCL refuses two
    ranks on a GPU, so PT_BENCH_REHEARSAL=1 puts the process group on gloo and stages the tile exchange through the host.  Everything else
    is the real thing -- HIP renderer, partition autotune over every candidate, batched submit / finish, barriers, totals, the one JSON
    line.  (The numbers of such a run mean nothing; the line says so.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--width", "640", "--height", "384"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "strong"
    tx = d["config"]["tile_exchange"]
    assert tx["autotune"]["chosen"] in tx["autotune"]["candidates"] and len(tx["autotune"]["seconds"]) == len(tx["autotune"]["candidates"])
    assert "rehearsal" in d["config"] and d["roofline"]["bound"] == "hbm"


def test_bench_on_two_gpus_over_rccl():
    """The real N > 1 run (needs a box with at least two GPUs; the build's own boxes have one, so this is skipped there): bench.py with two ranks
    over RCCL, measured through torch.distributed.gather, then -- after the result line -- the C-ABI's own exchange (pt_comm_init / pt_gather)
    brought up, compared bit for bit with the frames of the torch path and timed (bench.py cabi_post_check); and the same run with the C-ABI
    exchange as the measured path."""
    import json
    import os
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k, gather in enumerate(("torch", "cabi")):
        port = 29700 + (os.getpid() + k) % 200
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                            os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--gather", gather],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        tx = d["config"]["tile_exchange"]
        assert d["n_gpus"] == 2 and tx["ranks_seen"] == 2 and d["value"] > 0
        assert sum(r[1] for r in tx["rays_and_pixels_per_rank"]) == 1920 * 1080 * 12  # every pixel of every frame rendered exactly once
        if gather == "torch":
            check = [ln for ln in p.stderr.splitlines() if ln.startswith("[bench] cabi_check: ")]
            assert len(check) == 1, p.stderr[-3000:]
            c = json.loads(check[0][len("[bench] cabi_check: "):])
            assert c["ok"] and c["frames_bit_identical_to_torch_gather"], c
        else:
            assert "pt_gather" in tx["gather"]
