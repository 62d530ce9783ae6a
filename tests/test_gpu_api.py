"""Error behaviour and golden fixtures through the C-ABI on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_call_order_and_argument_errors(dxrs, host):
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    r = dxrs.Renderer()
    try:
        gs = dxrs.types.graphics_settings(64, 64)
        with pytest.raises(dxrs.PtError) as e:
            r.set_constants(gs); r.render()
        assert e.value.status == 4 and "pt_set_scene" in str(e.value)  # PT_ERR_STATE
        r.set_scene(spheres, materials, sd, build=False)
        with pytest.raises(dxrs.PtError) as e:
            r.render()
        assert e.value.status == 4 and "pt_build_accel" in str(e.value)
        r.build_accel()
        with pytest.raises(dxrs.PtError) as e:
            r.render()
        assert e.value.status == 4 and "pt_set_camera" in str(e.value)
        r.set_camera(host.camera(64, 64))
        img, _ = r.render()
        assert img.shape == (64, 64, 4)
        with pytest.raises(dxrs.PtError) as e:
            r.render(rect=(32, 32, 64, 8))
        assert e.value.status == 1
        for field, status in (("Denoiser", 5),):
            bad = dxrs.types.graphics_settings(64, 64); setattr(bad, field, 1)
            with pytest.raises(dxrs.PtError) as e:
                r.set_constants(bad)
            assert e.value.status == status
        for w, h, spp in ((0, 64, 1), (70000, 64, 1), (64, 64, 0)):
            with pytest.raises(dxrs.PtError) as e:
                r.set_constants(dxrs.types.graphics_settings(w, h, spp=spp))
            assert e.value.status == 1
        sd2 = host.scene(dxrs.host.SCENE_SMALL)[2]; sd2.EnvironmentLightTextureDescriptor = 3
        r.set_scene(spheres, materials, sd2); r.set_constants(gs)  # an environment map the texture table does not hold: refused at render time
        with pytest.raises(dxrs.PtError) as e:
            r.render()
        assert e.value.status == 4 and "EnvironmentLightTextureDescriptor" in str(e.value)
        bad_s = spheres.copy(); bad_s["r"][3] = 0
        with pytest.raises(dxrs.PtError) as e:
            r.set_scene(bad_s, materials, sd)
        assert e.value.status == 1 and "sphere 3" in str(e.value)
        with pytest.raises(dxrs.PtError):
            r.set_partition(2, 2)
        # a failed call leaves the context usable
        r.set_scene(spheres, materials, sd); r.set_constants(gs)
        assert r.render()[0].shape == (64, 64, 4)
    finally:
        r.close()


def test_golden_crops(dxrs, host, renderer):
    """GPU output == committed golden vectors (tests/golden_cases.py, written by tests/golden/make_golden.py from the CPU oracle):
    C1, C2, textured + environment map, direct illumination, cube environment, and the tone-mapped C2 crop"""
    import torch
    import golden_cases
    for c in golden_cases.cases(dxrs, host):
        renderer.set_scene(c["spheres"], c["materials"], c["sd"])
        renderer.set_textures(c["textures"])
        renderer.set_camera(c["cam"]); renderer.set_constants(c["gs"])
        img, _ = renderer.render(rect=c["rect"])
        gold = np.load(os.path.join(GOLD, c["file"]))
        assert np.array_equal(img.view(np.uint32)[..., :3], gold.view(np.uint32)[..., :3]), c["file"]
    renderer.set_textures(None)
    src, dst, params = golden_cases.tonemap_case(dxrs)
    hdr = torch.from_numpy(np.load(os.path.join(GOLD, src)).reshape(-1, 4)).cuda()
    ldr = torch.empty(hdr.shape[0], dtype=torch.int32, device="cuda")
    renderer.tonemap(hdr.data_ptr(), hdr.shape[0], params, ldr.data_ptr())
    renderer.synchronize()
    assert np.array_equal(ldr.cpu().numpy().view(np.uint32), np.load(os.path.join(GOLD, dst)).reshape(-1))


@pytest.mark.parametrize("flags", [1, 4, 5, 8, 9])  # NO_LDS_SCENE, HOST_LBVH, both, SPLIT_KERNELS, SPLIT + NO_LDS
def test_flag_variants_match_oracle(dxrs, host, oracle, flags):
    """The global-memory BVH path and the host-built LBVH give the same image as the default configuration / the oracle."""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 1920, 1080
    gs = dxrs.types.graphics_settings(w, h, frame_index=1, bounces=8, spp=2)
    cam = host.camera(w, h, jitter_index=1)
    rect = (860, 440, 192, 96)
    r = dxrs.Renderer(flags=flags)
    try:
        r.set_scene(spheres, materials, sd); r.set_camera(cam); r.set_constants(gs)
        img, st = r.render(rect)
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=8)
        assert st.rays == ost.rays and np.array_equal(img.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
    finally:
        r.close()


def test_large_scene_global_bvh_matches_oracle(dxrs, host, oracle):
    """BASELINE config C5 family (procedural spheres, BVH in global memory, 32-bit stack) on a size the brute-force
    oracle finishes in seconds: 100k spheres, small crop."""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_PROCEDURAL, seed=1, count=100000)
    w, h = 1920, 1080
    gs = dxrs.types.graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
    cam = host.camera(w, h, jitter_index=0)
    rect = (900, 560, 96, 48)
    r = dxrs.Renderer()
    try:
        info = r.set_scene(spheres, materials, sd)
        assert info.lds_resident == 0
        r.set_camera(cam); r.set_constants(gs)
        img, st = r.render(rect)
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=16)
        assert st.rays == ost.rays and np.array_equal(img.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
        full, _ = r.render()
        assert np.isfinite(full).all()
    finally:
        r.close()


def test_totals_accumulate_on_device(dxrs, host, renderer):
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(host.camera(128, 128))
    gs = dxrs.types.graphics_settings(128, 128, bounces=4)
    renderer.totals(reset=True)
    rays = 0
    for f in range(5):
        gs.FrameIndex = f
        renderer.set_constants(gs)
        _, st = renderer.render()
        rays += st.rays
    tot = renderer.totals(reset=True)
    assert tot.rays == rays and tot.pixels == 5 * 128 * 128 and tot.paths == tot.pixels
    assert renderer.totals().rays == 0


@pytest.mark.parametrize("flags", [0, 16])  # one frame at a time / PT_FLAG_TWO_FRAMES_IN_FLIGHT
def test_animated_scene_refit_matches_oracle(dxrs, host, oracle, flags):
    """the demo's motion (closed-form springs + Moon orbit) with a per-frame LBVH refit.  Every frame equals
    the oracle on that frame's sphere positions bit-for-bit (any valid BVH gives the brute-force answer), also when the
    spheres have drifted far from the positions the tree topology was built for."""
    import torch
    spheres0, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 1920, 1080
    rect = (800, 380, 320, 200)
    tstream = torch.cuda.Stream()
    r = dxrs.Renderer(stream=tstream.cuda_stream, flags=flags)
    try:
        with torch.cuda.stream(tstream):
            r.set_scene(spheres0, materials, sd)
            bufs = [torch.empty((rect[3], rect[2], 4), dtype=torch.float32, device="cuda") for _ in range(2)]
            times = [0.0, 0.05, 0.4, 0.75, 1.5, 2.25, 7.3, 31.0]
            frames = []
            for k, t in enumerate(times):
                sph = host.scene_at_time(0, t)
                gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=8, spp=1)
                r.update_spheres(sph)
                r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs)
                r.render_device(bufs[k % 2].data_ptr(), rect=rect)
                frames.append(bufs[k % 2].clone())
            torch.cuda.synchronize()
        for k, t in enumerate(times):
            sph = host.scene_at_time(0, t)
            gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=8, spp=1)
            ref, _ = oracle.render(sph, materials, sd, host.camera(w, h, jitter_index=k), gs, rect=rect, threads=8)
            assert np.array_equal(frames[k].cpu().numpy().view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"t={t}"
        # a count change is refused, and a fresh pt_set_scene returns every lane to the static scene
        with pytest.raises(dxrs.PtError):
            r.update_spheres(spheres0[:-1])
        r.set_scene(spheres0, materials, sd)
        r.set_camera(host.camera(w, h)); r.set_constants(dxrs.types.graphics_settings(w, h))
        a, _ = r.render(rect)
        ref, _ = oracle.render(spheres0, materials, sd, host.camera(w, h), dxrs.types.graphics_settings(w, h), rect=rect, threads=8)
        assert np.array_equal(a.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
    finally:
        r.close()


@pytest.mark.parametrize("n", [2, 3, 65, 1024, 1025, 1500])  # sphere counts incl. the ground: 1024 is the last single-launch size
@pytest.mark.parametrize("fused", ["1", "0"])
def test_moved_spheres_single_launch_refit(dxrs, host, oracle, n, fused, monkeypatch):
    """Row N2 for small scenes: pt_update_spheres leaves the new spheres in the lane's pinned staging buffer and pt_refit_accel's ONE kernel
    (refit_fused_small_kernel, n <= 1024) copies them, reduces the bounds, gathers and refits; PT_FUSED_REFIT=0 and larger scenes take the
    copy + separate kernels.  Frames of moved spheres equal the oracle on the moved positions either way, over three lanes, and a render call
    that follows an update WITHOUT a refit still sees the new spheres (here: unmoved ones, so the old boxes stay valid)."""
    import torch
    monkeypatch.setenv("PT_FUSED_REFIT", fused)
    spheres0, materials, sd = host.scene(dxrs.host.SCENE_PROCEDURAL, seed=5, count=n - 1)  # n - 1 spheres + the ground (last)
    assert len(spheres0) == n and spheres0["r"][-1] == 50.0
    w, h = 160, 96
    rng = np.random.default_rng(n)
    # into the camera's view: a 10 x 10 patch of the ground around the origin
    m = n - 1
    spheres0["r"][:m] = rng.uniform(0.1, 0.35, m).astype(np.float32)
    spheres0["cx"][:m] = rng.uniform(-5, 5, m).astype(np.float32)
    spheres0["cz"][:m] = rng.uniform(-5, 5, m).astype(np.float32)
    spheres0["cy"][:m] = (-0.1 + spheres0["r"][:m] + rng.uniform(0, 1, m)).astype(np.float32)
    tstream = torch.cuda.Stream()
    r = dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=3)
    try:
        with torch.cuda.stream(tstream):
            r.set_scene(spheres0, materials, sd)
            bufs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
            moved, frames = [], []
            for k in range(5):
                sph = spheres0.copy()
                if k != 3:  # frame 3: the original positions again, uploaded without a refit right after a refit to the same positions
                    sph["cx"][:m] += rng.uniform(-0.3, 0.3, m).astype(np.float32)
                    sph["cy"][:m] += rng.uniform(0.0, 0.5, m).astype(np.float32)
                    sph["r"][:m] *= rng.uniform(0.7, 1.2, m).astype(np.float32)
                    r.update_spheres(sph)
                else:
                    r.update_spheres(sph)                 # boxes refitted to the original positions ...
                    r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(dxrs.types.graphics_settings(w, h, frame_index=k, bounces=4, spp=1))
                    r.render_device(bufs[0].data_ptr())   # (three calls, so that the update below lands on the same lane)
                    r.render_device(bufs[1].data_ptr())
                    r.render_device(bufs[2].data_ptr())
                    r.update_spheres(sph, refit=False)    # ... and the same spheres again, no refit: the render call uploads them
                moved.append(sph)
                gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=4, spp=1)
                r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs)
                r.render_device(bufs[k % 3].data_ptr())
                frames.append(bufs[k % 3].clone())
            torch.cuda.synchronize()
        for k in range(5):
            gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=4, spp=1)
            ref, _ = oracle.render(moved[k], materials, sd, host.camera(w, h, jitter_index=k), gs, threads=8)
            assert np.array_equal(frames[k].cpu().numpy().view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"frame {k}"
    finally:
        r.close()


@pytest.mark.parametrize("w,h,spp,bounces,lanes", [(1920, 1080, 1, 8, 1), (1920, 1080, 1, 8, 3), (1280, 720, 4, 6, 2), (333, 211, 1, 3, 1), (640, 360, 2, 1, 1)])
def test_segmented_and_dense_hand_over_agree(dxrs, host, w, h, spp, bounces, lanes, monkeypatch):
    """the queue between the primary pass and the looping pass is segmented per workgroup by default (no barrier, no global
    atomic per batch), dense with PT_SEG=0, and consumed by the primary pass itself with PT_FUSE_LOOP=1 (the default
    for small frames): same frames bit for bit, same ray counts, same queue sizes -- over several frames in flight, so that the
    per-lane segment counters and work cursors are reused"""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    import torch
    results = {}
    for seg, fuse in (("1", "0"), ("0", "0"), ("1", "1")):  # segmented, dense, and the single-launch form (the primary pass finishes its own segments)
        monkeypatch.setenv("PT_SEG", seg)
        monkeypatch.setenv("PT_FUSE_LOOP", fuse)
        r = dxrs.Renderer(frames_in_flight=lanes)
        try:
            r.set_scene(spheres, materials, sd)
            gs = dxrs.types.graphics_settings(w, h, bounces=bounces, spp=spp)
            bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(max(lanes, 1))]
            frames, rays = [], []
            for k in range(2 * max(lanes, 1) + 1):
                gs.FrameIndex = k
                r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs)
                r.render_device(bufs[k % len(bufs)].data_ptr())
                if (k + 1) % len(bufs) == 0 or k == 2 * max(lanes, 1):
                    r.synchronize()
                    frames.append(bufs[k % len(bufs)].cpu().numpy().copy())
            tot = r.totals(reset=True)
            results[seg + fuse] = (frames, int(tot.rays), int(tot.paths), list(r.queue_sizes())[:3])
        finally:
            r.close()
    a = results["10"]
    assert a[3][1] > 0
    for b in (results["00"], results["11"]):
        assert a[1] == b[1] and a[2] == b[2] and a[3][:2] == b[3][:2]
        for fa_, fb_ in zip(a[0], b[0]):
            assert np.array_equal(fa_.view(np.uint32), fb_.view(np.uint32))


def test_primary_beam_cache_follows_the_view(dxrs, host, oracle, renderer):
    """Primary-beam lists (DESIGN.md "Primary beams") are built when a view rests and must never outlive it: a camera move, a
    different rect, a new scene or moved spheres each fall back to per-ray traversal until the new view has rested -- every
    frame bit-identical to the oracle whichever path it took."""
    from util import count_mismatch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 320, 200
    gs = dxrs.types.graphics_settings(w, h, frame_index=3, bounces=4, spp=1)
    cam_a = host.camera(w, h, jitter_index=1)
    cam_b = host.camera(w, h, position=(1.5, 0.5, -13.0), jitter_index=2)
    renderer.set_scene(spheres, materials, sd); renderer.set_constants(gs)
    ref = {}

    def check(cam, name, expect, rect=None, scene=(spheres, materials, sd)):
        renderer.set_camera(cam)
        img, st = renderer.render(rect)
        key = (name, rect, id(scene[0]))
        if key not in ref:
            ref[key] = oracle.render(scene[0], scene[1], scene[2], cam, gs, rect=rect, threads=8)
        assert bool(st.beams_used) == expect, (name, st.beams_used)
        assert st.rays == ref[key][1].rays and count_mismatch(img, ref[key][0]) == 0, name

    check(cam_a, "a", False); check(cam_a, "a", False); check(cam_a, "a", True)
    check(cam_b, "b", False)                       # the camera moved: the lists are for another view
    check(cam_a, "a", True)                        # back: still the cached view (nothing rebuilt in between)
    check(cam_b, "b", False); check(cam_b, "b", False); check(cam_b, "b", True)  # b rested: rebuilt for b
    check(cam_b, "b", False, rect=(40, 30, 100, 64))   # another rect = another slot -> pixel map
    check(cam_b, "b", False, rect=(40, 30, 100, 64)); check(cam_b, "b", True, rect=(40, 30, 100, 64))
    # the same view with another jitter and frame index keeps the lists (they are a pixel wider than their blocks)
    gs.FrameIndex = 9
    renderer.set_constants(gs)
    ref.clear()
    cam_b2 = host.camera(w, h, position=(1.5, 0.5, -13.0), jitter_index=5)
    check(cam_b2, "b2", True, rect=(40, 30, 100, 64))
    # moved spheres (animation: per-lane private scene copies) never use lists built from the master scene
    moved = host.scene_at_time(0, 1.25)
    renderer.update_spheres(moved)
    scene2 = (moved, materials, sd)
    check(cam_b2, "b2", False, rect=(40, 30, 100, 64), scene=scene2)
    renderer.update_spheres(moved)
    check(cam_b2, "b2", False, rect=(40, 30, 100, 64), scene=scene2)
    # a new scene
    small = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    renderer.set_scene(*small)
    check(cam_b2, "b2", False, scene=small); check(cam_b2, "b2", False, scene=small); check(cam_b2, "b2", True, scene=small)


def test_rect_bounds_do_not_wrap(dxrs, host, renderer):
    """x + w computed in 32 bits used to wrap (x = 0xFFFFFFFF, w = 2 passed the check)"""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(host.camera(64, 48)); renderer.set_constants(dxrs.types.graphics_settings(64, 48, bounces=1))
    for rect in ((0xFFFFFFFF, 0, 2, 1), (0, 0xFFFFFFFF, 1, 2), (63, 0, 2, 1), (0, 47, 1, 2), (64, 0, 1, 1), (0, 0, 0, 1)):
        with pytest.raises(dxrs.PtError):
            renderer.render(rect)
    img, _ = renderer.render((63, 47, 1, 1))
    assert img.shape == (1, 1, 4)


def test_same_output_buffer_on_consecutive_frames_in_flight(dxrs, host):
    """with N frames in flight the caller is meant to rotate over N output buffers; handing the SAME buffer to
    consecutive calls used to let two lanes write it at once.  The context now orders such frames itself: every frame read back
    after its call must be that frame, bit for bit."""
    import torch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 640, 360
    ts = torch.cuda.Stream()  # the caller's stream: the context orders its frames against it, the read-backs below run on it
    ref = dxrs.Renderer(frames_in_flight=1)
    r = dxrs.Renderer(stream=ts.cuda_stream, frames_in_flight=3)
    try:
        gs = dxrs.types.graphics_settings(w, h, bounces=6, spp=1)
        for x in (ref, r):
            x.set_scene(spheres, materials, sd)
        buf = torch.empty((h * w, 4), dtype=torch.float32, device="cuda")
        snap = [torch.empty_like(buf) for _ in range(7)]
        want = []
        for k in range(7):
            gs.FrameIndex = k
            cam = host.camera(w, h, jitter_index=k)
            ref.set_camera(cam); ref.set_constants(gs)
            want.append(ref.render()[0].reshape(h * w, 4))
            r.set_camera(cam); r.set_constants(gs)
            r.render_device(buf.data_ptr())       # the same buffer every time, no synchronisation in between
            with torch.cuda.stream(ts):
                snap[k].copy_(buf, non_blocking=True)  # the consumer, on the caller's stream
        r.synchronize()
        torch.cuda.synchronize()
        for k in range(7):
            assert np.array_equal(snap[k].cpu().numpy().view(np.uint32), want[k].view(np.uint32)), f"frame {k}"
    finally:
        r.close(); ref.close()


def test_empty_scene(dxrs, host, oracle):
    """pt_set_scene with n = 0 (the reference's TLAS may hold no instance): every pixel is the environment, bit-identical to the oracle's
    empty scene -- plain, with several samples, in tiles, with a lat-long environment map, with frames in flight; moving "all zero"
    spheres is a no-op, moving one is refused; a real scene afterwards renders as ever."""
    import torch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 200, 120
    tstream = torch.cuda.Stream()
    r = dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=3)
    try:
        with torch.cuda.stream(tstream):
            info = r.set_scene(spheres[:0], materials[:0], sd)
            assert info.leaf_count == 0 and info.node_count == 0
            for k, (spp, bounces) in enumerate(((1, 8), (3, 4), (1, 0))):
                gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=bounces, spp=spp)
                cam = host.camera(w, h, jitter_index=k)
                r.set_camera(cam); r.set_constants(gs)
                img, st = r.render()
                ref, ost = oracle.render(spheres[:0], materials[:0], sd, cam, gs, threads=4)
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.rays == ost.rays == w * h
            r.update_spheres(spheres[:0])  # nothing to move
            with pytest.raises(dxrs.PtError):
                r.update_spheres(spheres[:1])
            # tiles: rank 1 of 3 renders its tiles of the empty scene
            from dxrs_amd import tiles
            gs = dxrs.types.graphics_settings(w, h, frame_index=5, bounces=8, spp=1)
            cam = host.camera(w, h, jitter_index=5)
            r.set_camera(cam); r.set_constants(gs)
            full, _ = r.render()
            r.set_partition(1, 3)
            packed = torch.empty((r.tiles_count(1) * 1024, 4), dtype=torch.float32, device="cuda")  # (no fill: see PT_FLAG_TWO_FRAMES_IN_FLIGHT)
            r.render_tiles(packed.data_ptr())
            torch.cuda.synchronize()
            want = tiles.pack_range(full, 1, 1, 3)
            assert np.array_equal(packed.cpu().numpy().reshape(want.shape).view(np.uint32), want.view(np.uint32))
            r.set_partition(0, 1)
            # lit by the lat-long environment map (texture table without objects)
            tex, sd_env = host.demo_textures(0, 0.0, textured=False, environment_map=True, return_scene_data=True)
            tex_empty = type(tex)(0)
            tex_empty.images = tex.images
            r.set_scene(spheres[:0], materials[:0], sd_env)
            r.set_textures(tex_empty)
            img, _ = r.render()
            ref, _ = oracle.render(spheres[:0], materials[:0], sd_env, cam, gs, threads=4, textures=tex_empty)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            # and back to a real scene
            r.set_scene(spheres, materials, sd)
            r.set_camera(cam); r.set_constants(gs)
            rect = (60, 40, 64, 48)
            a, _ = r.render(rect)
            ref, _ = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=8)
            assert np.array_equal(a.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
    finally:
        r.close()


def test_primary_beam_lists_follow_a_moving_camera(dxrs, host, oracle, renderer):
    """A camera that moves without turning (App::Update's translation, Source/App.cpp:531-553) keeps primary-beam lists: they are built around a
    position a few frames ahead with a slack of a few frames' travel (Beam::slack: every plane moved outwards by that distance), renewed on the
    side stream before they run out.  Every frame is bit-identical to the oracle whichever way its primaries were found, most frames of a
    steady motion use lists, a turn or a jump falls back to traversal until the motion is steady again."""
    import math
    from util import count_mismatch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h, rect = 1920, 1080, (832, 476, 192, 96)
    gs = dxrs.types.graphics_settings(w, h, frame_index=0, bounces=4, spp=1)
    renderer.set_scene(spheres, materials, sd)
    used = []

    def frame(k, position, look_at=None):
        gs.FrameIndex = k
        cam = host.camera(w, h, position=position, look_at=look_at, jitter_index=k % 8)
        renderer.set_constants(gs); renderer.set_camera(cam)
        img, st = renderer.render(rect)
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=8)
        assert st.rays == ost.rays and count_mismatch(img, ref) == 0, (k, position)
        used.append(bool(st.beams_used))

    path = lambda k: (0.6 * math.sin(0.01 * k), 0.05 * math.sin(0.013 * k), -15.0 + 0.4 * math.cos(0.01 * k))  # bench.py --moving-camera
    for k in range(20):
        frame(k, path(k))
    assert not used[0] and not used[1]          # nothing to go by yet; the first build starts on the second frame
    assert sum(used[2:]) >= 14, used            # steady motion: lists nearly always
    n0 = len(used)
    frame(20, path(20), look_at=(0.5, 0.0, 0.0))   # a turn: another orientation
    frame(21, path(21), look_at=(0.5, 0.0, 0.0))
    assert not used[n0]
    frame(22, (3.0, 1.0, -12.0))                   # a jump: no lists can be worth their slack
    assert not used[-1]
    for k in range(23, 31):
        frame(k, (3.0 + 0.002 * (k - 22), 1.0, -12.0))  # steady again, slowly
    assert sum(used[-5:]) >= 3, used
    # a resting view still gets exact lists (no slack) on its third frame
    for k in range(31, 34):
        frame(k, (3.1, 1.0, -12.0))
    assert used[-1]


def test_moving_camera_frames_in_flight_match_per_ray_traversal(dxrs, host, renderer_no_beams):
    """Three frames in flight over a moving camera: lists are taken when a build has finished, never waited for -- whatever the timing, the frames
    equal those of a context without primary beams, bit for bit."""
    import math
    torch = pytest.importorskip("torch")
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h, n = 640, 360, 60
    gs = dxrs.types.graphics_settings(w, h, frame_index=0, bounces=3, spp=1)
    cams = [host.camera(w, h, position=(0.3 * math.sin(0.02 * k), 0.0, -15.0 + 0.002 * k), jitter_index=k % 8) for k in range(n)]
    out = {}
    r3 = dxrs.Renderer(device=0, frames_in_flight=3)
    try:
        for name, r in (("beams", r3), ("plain", renderer_no_beams)):
            r.set_scene(spheres, materials, sd)
            bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda:0") for _ in range(n)]
            r.totals(reset=True)
            for k in range(n):
                gs.FrameIndex = k
                r.set_constants(gs); r.set_camera(cams[k])
                r.render_device(bufs[k].data_ptr())
            r.synchronize()
            tot = r.totals(reset=True)
            out[name] = ([b.cpu().numpy().view(np.uint32) for b in bufs], int(tot.rays), int(tot.beams_used))
    finally:
        r3.close()
    assert out["plain"][2] == 0 and out["beams"][2] > n // 3, (out["plain"][2], out["beams"][2])
    assert out["beams"][1] == out["plain"][1]
    for k in range(n):
        assert np.array_equal(out["beams"][0][k], out["plain"][0][k]), k


def test_primary_beam_lists_follow_a_turning_camera(dxrs, host, oracle, renderer):
    """A camera that turns slowly (App::Update's mouse look) keeps primary-beam lists too: they are built for the orientation extrapolated a few
    frames ahead, every block's outline widened by the pixels a ray's crossing of the image can move within the turn the lists are to cover
    (make_beam margin_px; pt_api.hip beam_cache_lookup bounds the displacement at the image corner).  Whole frames against the oracle -- the
    corners are where the bound is tightest -- while the camera yaws, pitches, rolls about a moving position, and turns too fast for any list."""
    import math
    from util import count_mismatch
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 320, 200
    gs = dxrs.types.graphics_settings(w, h, frame_index=0, bounces=3, spp=1)
    renderer.set_scene(spheres, materials, sd)
    used = []

    def frame(k, position, look_at):
        gs.FrameIndex = k
        cam = host.camera(w, h, position=position, look_at=look_at, jitter_index=k % 8)
        renderer.set_constants(gs); renderer.set_camera(cam)
        img, st = renderer.render()
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
        assert st.rays == ost.rays and count_mismatch(img, ref) == 0, (k, position, look_at)
        used.append(bool(st.beams_used))

    for k in range(16):                                    # a slow yaw about a fixed position
        frame(k, (0.0, 1.0, -14.0), (0.004 * k, 0.0, 0.0))
    assert sum(used[3:]) >= 10, used
    n0 = len(used)
    for k in range(16, 32):                                # yaw + pitch while the camera also travels
        j = k - 16
        frame(k, (0.003 * j, 1.0 + 0.002 * j, -14.0 + 0.004 * j), (0.06 + 0.003 * j, 0.002 * j, 0.0))
    assert sum(used[n0 + 4:]) >= 8, used
    n1 = len(used)
    for k in range(32, 38):                                # a fast turn: no list can be worth its margin
        frame(k, (0.05, 1.03, -13.94), (0.1 + 0.5 * (k - 31), 0.03, 0.0))
    assert not any(used[n1 + 1:]), used
