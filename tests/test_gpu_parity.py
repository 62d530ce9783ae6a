"""GPU parity proper: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.
Bar (BASELINE.json north_star): per-pixel radiance within 1e-4 relative L2; the integer RNG stream bit-exact
(implied by bit-identical ray counts and images).  In practice the two agree bit-for-bit."""
import numpy as np
import pytest

from util import count_mismatch, rel_l2, render_rested

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative L2, stated by north_star


def _render_both(dxrs, host, oracle, renderer, scene, w, h, bounces, spp, rect=None, frame=0, rr=True, jitter_index=0):
    spheres, materials, sd = scene
    gs = dxrs.types.graphics_settings(w, h, frame_index=frame, bounces=bounces, spp=spp, rr=rr)
    cam = host.camera(w, h, jitter_index=jitter_index)
    renderer.set_scene(spheres, materials, sd)
    renderer.set_camera(cam)
    renderer.set_constants(gs)
    # three frames of the view: per-ray BVH traversal for the primaries first, primary-beam candidate lists on the third
    img, stats = render_rested(renderer, rect, expect_beams=len(spheres) > 1)
    ref, ostats = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=8)
    return img, stats, ref, ostats


def test_c1_full_frame(dxrs, host, oracle, renderer):
    """BASELINE config C1: 16 spheres, 256x256, 1 spp, 4 bounces."""
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 256, 256, 4, 1)
    assert stats.rays == ostats.rays
    assert stats.paths == ostats.paths
    assert rel_l2(img, ref) <= TOL
    assert count_mismatch(img, ref) == 0


@pytest.mark.parametrize("rect", [(832, 476, 256, 128), (0, 0, 128, 64), (1792, 1016, 128, 64), (900, 300, 97, 53)])
def test_c2_crops(dxrs, host, oracle, renderer, rect):
    """BASELINE config C2 (demo scene, 1080p, 1 spp, 8 bounces) on crops the oracle finishes in seconds."""
    scene = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 1920, 1080, 8, 1, rect=rect)
    assert stats.rays == ostats.rays
    assert rel_l2(img, ref) <= TOL
    assert count_mismatch(img, ref) == 0


@pytest.mark.parametrize("spp,bounces", [(2, 4), (4, 8), (16, 8)])
def test_multi_sample_regeneration(dxrs, host, oracle, renderer, spp, bounces):
    """spp > 1: samples of a pixel share one sequential RNG stream (Raytracing.hlsl:108,191)."""
    scene = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 3840, 2160, bounces, spp, rect=(1800, 1000, 160, 96))
    assert stats.rays == ostats.rays
    assert rel_l2(img, ref) <= TOL
    assert count_mismatch(img, ref) == 0


def test_c4_crop_max_spp_and_bounces(dxrs, host, oracle, renderer):
    """BASELINE config C4: 3840x2160, 64 spp, 16 bounces (the UI maximum, Source/MyAppData.h:183-188) -- a 64x48 crop the oracle
    finishes in seconds (~3 M rays): 64 regenerated samples per pixel share one RNG stream, the sample counter and flags of the
    48-byte ray record are exercised up to sample 63, and the looping pass runs up to 64 x 17 iterations per pixel."""
    import os
    scene = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    rect = (1888, 1016, 64, 48)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 3840, 2160, 16, 64, rect=rect)
    assert stats.rays == ostats.rays and stats.paths == ostats.paths == 64 * 48 * 64
    assert rel_l2(img, ref) <= TOL
    assert count_mismatch(img, ref) == 0
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4_crop_1888_1016_64x48.npy"))
    assert count_mismatch(img, gold) == 0
    # the same pixels out of a larger rect whose origin is not 8-aligned (another slot -> pixel mapping, same RNG keys)
    big = (1861, 1003, 131, 75)
    renderer.set_constants(dxrs.types.graphics_settings(3840, 2160, frame_index=0, bounces=16, spp=64))
    img2, _ = renderer.render(big)
    sub = img2[rect[1] - big[1]: rect[1] - big[1] + rect[3], rect[0] - big[0]: rect[0] - big[0] + rect[2]]
    assert count_mismatch(sub, gold) == 0


def test_frames_and_jitter(dxrs, host, oracle, renderer):
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    for frame in (1, 5, 1234567):
        img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 200, 120, 8, 1, frame=frame, jitter_index=frame)
        assert stats.rays == ostats.rays
        assert count_mismatch(img, ref) == 0


def test_rr_off_and_zero_bounces(dxrs, host, oracle, renderer):
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    for bounces, rr in ((8, False), (0, True), (1, True)):
        img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 160, 90, bounces, 1, rr=rr)
        assert stats.rays == ostats.rays
        assert count_mismatch(img, ref) == 0


@pytest.mark.parametrize("w,h,rect", [(1, 1, None), (7, 5, None), (9, 17, (8, 16, 1, 1)), (100, 3, (93, 0, 7, 3)), (33, 65, None)])
def test_ragged_and_tiny_frames(dxrs, host, oracle, renderer, w, h, rect):
    """Frame sizes that are not multiples of the 8x8 wave block, 1x1 frames and 1x1 rects at the far corner."""
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, w, h, 4, 2, rect=rect)
    assert img.shape == ref.shape and stats.rays == ostats.rays and stats.pixels == img.shape[0] * img.shape[1]
    assert count_mismatch(img, ref) == 0


@pytest.mark.parametrize("w,h", [(65535, 1), (1, 65535), (65535, 3), (2, 40000)])
def test_maximum_extents(dxrs, host, oracle, renderer, w, h):
    """The largest RenderSize the interface accepts in one dimension (65535: the RNG seed packs (x << 16) | y), as slivers the oracle finishes
    in seconds: whole frames and, through the tile path, one rank's share of them."""
    import torch
    from dxrs_amd import tiles
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, w, h, 3, 1)
    assert img.shape == ref.shape == (h, w, 4) and stats.rays == ostats.rays and stats.pixels == w * h
    assert count_mismatch(img, ref) == 0
    try:
        renderer.set_partition(1, 3)
        n_tiles = renderer.tiles_count(1)
        packed = torch.empty((n_tiles * 1024, 4), dtype=torch.float32, device="cuda")
        renderer.render_tiles(packed.data_ptr())
        torch.cuda.synchronize()
        want = tiles.pack_range(ref, 1, 1, 3)
        assert want.shape[0] == n_tiles
        assert np.array_equal(packed.cpu().numpy().reshape(want.shape).view(np.uint32)[..., :3], want.view(np.uint32)[..., :3])
    finally:
        renderer.set_partition(0, 1)


@pytest.mark.parametrize("w,h,spp,bounces,rr", [(4, 3, 65535, 2, True), (32, 16, 1, 250, False), (8, 8, 300, 250, False)])
def test_maximum_counts(dxrs, host, oracle, renderer, w, h, spp, bounces, rr):
    """The largest SamplesPerPixel (65535) and Bounces (250) the interface accepts: the sample counter and the bounce counter of the ray
    record at the ends of their ranges, the looping pass at its longest."""
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, w, h, bounces, spp, rr=rr)
    assert stats.rays == ostats.rays
    assert count_mismatch(img, ref) == 0


@pytest.mark.parametrize("near,far", [(0.0, float("inf")), (0.0, 3.0e38), (5.0, 14.0), (20.0, 10.0)])
@pytest.mark.parametrize("empty", [False, True])
def test_depth_range_extremes(dxrs, host, oracle, renderer, near, far, empty):
    """NearDepth / FarDepth bound the primary rays only (Camera.hlsli:27-41): no bound at all (0 .. inf -- also the case in which the
    stand-in sphere of an empty scene must stay unhittable without the help of tmax), a slab that cuts the scene, and an empty range."""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    if empty:
        spheres, materials = spheres[:0], materials[:0]
    w, h = 96, 64
    gs = dxrs.types.graphics_settings(w, h, frame_index=3, bounces=4, spp=2)
    cam = host.camera(w, h, jitter_index=3)
    cam.NearDepth, cam.FarDepth = near, far
    renderer.set_scene(spheres, materials, sd)
    renderer.set_camera(cam); renderer.set_constants(gs)
    img, stats = render_rested(renderer, None)
    ref, ostats = oracle.render(spheres, materials, sd, cam, gs, threads=8)
    assert stats.rays == ostats.rays and np.isfinite(img).all()
    assert count_mismatch(img, ref) == 0


def test_more_ranks_than_tiles(dxrs, host, renderer):
    """A rank that owns no tile renders nothing and reports zero work."""
    import torch
    scene = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    renderer.set_scene(*scene); renderer.set_camera(host.camera(40, 40)); renderer.set_constants(dxrs.types.graphics_settings(40, 40, bounces=2))
    try:
        renderer.set_partition(5, 8)  # 2 x 2 tiles: ranks 4..7 own nothing
        assert renderer.tiles_count(5) == 0 and renderer.tiles_count(3) == 1
        buf = torch.zeros((1024, 4), dtype=torch.float32, device="cuda")
        st = renderer.render_tiles(buf.data_ptr(), want_stats=True)
        assert st.rays == 0 and float(buf.abs().sum()) == 0.0
    finally:
        renderer.set_partition(0, 1)


def test_deep_paths(dxrs, host, oracle, renderer):
    """Many bounces with Russian roulette off: long transmission chains inside the glass spheres (looping pass)."""
    scene = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    img, stats, ref, ostats = _render_both(dxrs, host, oracle, renderer, scene, 1920, 1080, 40, 1, rect=(900, 420, 128, 96), rr=False)
    assert stats.rays == ostats.rays and count_mismatch(img, ref) == 0


@pytest.mark.parametrize("case", ["c1", "c2", "c2_spp4"])
def test_primary_rays_through_the_bvh_without_beams(dxrs, host, oracle, renderer, renderer_no_beams, case):
    """PT_BEAMS=0: the primary rays take the per-ray BVH traversal (what every frame did in round 1); same bits as the oracle and as the
    beam-list frames of the default context."""
    kind, w, h, bounces, spp, rect = {"c1": (dxrs.host.SCENE_SMALL, 256, 256, 4, 1, None), "c2": (dxrs.host.SCENE_DEMO, 1920, 1080, 8, 1, (832, 476, 256, 128)),
                                      "c2_spp4": (dxrs.host.SCENE_DEMO, 1280, 720, 6, 4, (500, 300, 160, 96))}[case]
    spheres, materials, sd = host.scene(kind, seed=0)
    gs = dxrs.types.graphics_settings(w, h, frame_index=4, bounces=bounces, spp=spp)
    cam = host.camera(w, h, jitter_index=4)
    imgs = []
    for r in (renderer_no_beams, renderer):
        r.set_scene(spheres, materials, sd); r.set_camera(cam); r.set_constants(gs)
        if r is renderer:
            img, st = render_rested(r, rect, expect_beams=True)
        else:
            r.render(rect); r.render(rect)
            img, st = r.render(rect)
            assert st.beams_used == 0
        imgs.append((img, st.rays))
    ref, ost = oracle.render(spheres, materials, sd, cam, gs, rect=rect, threads=8)
    for img, rays in imgs:
        assert rays == ost.rays and count_mismatch(img, ref) == 0
