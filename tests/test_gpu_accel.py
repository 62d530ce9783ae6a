"""LBVH on the device: traversal result == brute force (GPU kernel and CPU oracle) bit-for-bit, structural invariants of
the tree the kernels actually traverse, for LDS-resident and global-memory BVHs."""
import ctypes as C

import numpy as np
import pytest

from test_abi import check_lbvh

pytestmark = pytest.mark.gpu


def make_rays(spheres, n, seed):
    """half the rays aimed at (jittered) spheres from points in the scene bounds, half random; unit directions"""
    rng = np.random.default_rng(seed)
    c = np.stack([spheres["cx"], spheres["cy"], spheres["cz"]], 1).astype(np.float64)
    r = spheres["r"].astype(np.float64)
    lo, hi = (c - r[:, None]).min(0), (c + r[:, None]).max(0)
    lo = np.maximum(lo, -60); hi = np.minimum(hi, 60)  # keep most origins near the action (the ground sphere is huge)
    o = rng.uniform(lo, hi, (n, 3))
    pick = rng.integers(0, len(spheres), n)
    target = c[pick] + rng.normal(size=(n, 3)) * r[pick, None] * 0.7
    d = np.where((np.arange(n) % 2 == 0)[:, None], target - o, rng.normal(size=(n, 3)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # rays starting ON a sphere surface (the secondary-ray case) for a quarter of them
    k = np.arange(n) % 4 == 1
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    o[k] = c[pick[k]] + nrm[k] * r[pick[k], None] * (1 + 2 ** -14)
    d32 = d.astype(np.float32)
    # axis-parallel and plane-parallel rays: direction components that are exactly +0 / -0 (the slab test divides by
    # them; a frame of odd width with zero jitter has a whole column of such rays)
    ax = np.arange(n) % 16 == 3
    zero_mask = rng.integers(1, 7, n)  # bit c set -> component c is zeroed (never all three)
    for c in range(3):
        z = ax & ((zero_mask >> c) & 1).astype(bool)
        d32[z, c] = np.where(rng.random(int(z.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
    d32 /= np.linalg.norm(d32.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    return o.astype(np.float32), d32


SCENES = [("small", 0), ("demo", 0), ("procedural", 3000), ("procedural", 40000), ("procedural", 200000)]


@pytest.mark.parametrize("name,count", SCENES)
@pytest.mark.parametrize("flags", [0, 1])  # 1 = PT_FLAG_NO_LDS_SCENE
def test_bvh_equals_brute_force(dxrs, host, oracle, name, count, flags):
    kind = {"small": dxrs.host.SCENE_SMALL, "demo": dxrs.host.SCENE_DEMO, "procedural": dxrs.host.SCENE_PROCEDURAL}[name]
    spheres, materials, sd = host.scene(kind, seed=1, count=count)
    r = dxrs.Renderer(flags=flags)
    try:
        info = r.set_scene(spheres, materials, sd)
        assert info.leaf_count == len(spheres) and info.node_count == len(spheres) - 1
        if flags:
            assert info.lds_resident == 0
        n = 200000 if len(spheres) < 50000 else 20000
        o, d = make_rays(spheres, n, seed=len(spheres))
        for tmin in (0.0, 0.5):
            t_bvh, id_bvh = r.trace_rays(o, d, tmin=tmin, use_bvh=True)
            t_bf, id_bf = r.trace_rays(o, d, tmin=tmin, use_bvh=False)
            assert np.array_equal(id_bvh, id_bf)
            assert np.array_equal(t_bvh.view(np.uint32), t_bf.view(np.uint32))
            assert (id_bvh != 0xFFFFFFFF).mean() > 0.3
        # CPU oracle on a subset: same closest hit, bit-exact t
        sub = slice(0, 2000 if len(spheres) < 50000 else 300)
        lib = oracle.lib
        t_bvh, id_bvh = r.trace_rays(o[sub], d[sub], tmin=0.0, use_bvh=True)
        for i in range(len(t_bvh)):
            best, best_id = np.float32(np.inf), 0xFFFFFFFF
            tt = C.c_float()
            oi, di = np.ascontiguousarray(o[sub][i]), np.ascontiguousarray(d[sub][i])
            cand = np.nonzero(candidates(spheres, oi, di))[0]
            for sid in cand:
                if lib.oracle_intersect_sphere(oi.ctypes.data_as(C.POINTER(C.c_float)), di.ctypes.data_as(C.POINTER(C.c_float)),
                                               C.c_float(0.0), C.c_float(best), spheres[sid:sid + 1].ctypes.data, C.byref(tt)):
                    best, best_id = np.float32(tt.value), sid
            assert best_id == id_bvh[i] and (best_id == 0xFFFFFFFF or best == t_bvh[i])
    finally:
        r.close()


@pytest.mark.parametrize("kind", ["concentric", "line", "identical", "far_from_origin", "two_clusters"])
@pytest.mark.parametrize("n", [2, 500, 6000])
@pytest.mark.parametrize("flags", [0, 64, 1])  # default builder (SAH up to 4096 spheres) / PT_FLAG_FAST_BUILD (device LBVH) / PT_FLAG_NO_LDS_SCENE
def test_degenerate_layouts_bvh_equals_brute_force(dxrs, host, kind, n, flags):
    """Sphere layouts that degenerate the builders' inputs: one centre for all spheres (every Morton key equal, zero-extent centroid
    bounds), centres on a line (two axes without extent), n copies of one sphere (every key and every box equal: ties by index), a cluster
    far from the origin (padding and quantisation at large coordinates), two clusters a million units apart (nearly all Morton cells
    empty).  Whatever tree comes out, the closest hit must be the brute-force one, bit for bit, ties to the lowest id."""
    rng = np.random.default_rng(n * 31 + len(kind))
    s = np.zeros(n, dtype=dxrs.SPHERE_DTYPE)
    if kind == "concentric":
        s["cx"], s["cy"], s["cz"] = 0.5, -0.25, 2.0
        s["r"] = rng.uniform(0.1, 5.0, n)
    elif kind == "line":
        s["cx"] = rng.uniform(-50, 50, n); s["cy"], s["cz"] = 1.0, -3.0
        s["r"] = rng.uniform(0.05, 0.6, n)
    elif kind == "identical":
        s["cx"], s["cy"], s["cz"], s["r"] = 1.0, 2.0, 3.0, 0.75
    elif kind == "far_from_origin":
        s["cx"] = 1.0e5 + rng.uniform(-3, 3, n); s["cy"] = -2.0e5 + rng.uniform(-3, 3, n); s["cz"] = 3.0e4 + rng.uniform(-3, 3, n)
        s["r"] = rng.uniform(0.05, 0.5, n)
    else:
        half = n // 2
        s["cx"][:half] = rng.uniform(-2, 2, half); s["cx"][half:] = 1.0e6 + rng.uniform(-2, 2, n - half)
        s["cy"] = rng.uniform(-2, 2, n); s["cz"] = rng.uniform(-2, 2, n)
        s["r"] = rng.uniform(0.05, 0.5, n)
    m = dxrs.types.default_material(n)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    r = dxrs.Renderer(flags=flags)
    try:
        info = r.set_scene(s, m, sd)
        assert info.leaf_count == n and info.node_count == n - 1
        # rays from near a randomly picked sphere towards (a jittered point of) another one, half of them; random directions for the rest
        nr = 40000
        c = np.stack([s["cx"], s["cy"], s["cz"]], 1).astype(np.float64)
        a_, b_ = rng.integers(0, n, nr), rng.integers(0, n, nr)
        o = c[a_] + rng.normal(size=(nr, 3)) * 4.0
        tgt = c[b_] + rng.normal(size=(nr, 3)) * s["r"][b_, None] * 0.7
        d = np.where((np.arange(nr) % 2 == 0)[:, None], tgt - o, rng.normal(size=(nr, 3)))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o, d = o.astype(np.float32), d.astype(np.float32)
        d /= np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
        for tmin in (0.0, 0.3):
            t_bvh, id_bvh = r.trace_rays(o, d, tmin=tmin, use_bvh=True)
            t_bf, id_bf = r.trace_rays(o, d, tmin=tmin, use_bvh=False)
            assert np.array_equal(id_bvh, id_bf) and np.array_equal(t_bvh.view(np.uint32), t_bf.view(np.uint32))
            assert (id_bvh != 0xFFFFFFFF).mean() > 0.2
    finally:
        r.close()


def candidates(spheres, o, d):
    """cheap float64 prefilter so the per-ray oracle loop only visits spheres the ray passes near (margin 1e-3 r)"""
    c = np.stack([spheres["cx"], spheres["cy"], spheres["cz"]], 1).astype(np.float64) - o.astype(np.float64)
    d = d.astype(np.float64); d = d / np.linalg.norm(d)  # a float32 "unit" vector is off by 6e-8: matters at b ~ 100
    b = c @ d
    dist2 = (c * c).sum(1) - b * b
    rr = spheres["r"].astype(np.float64) * 1.01 + 1e-3
    return dist2 <= rr * rr


@pytest.mark.parametrize("name,count,fast_build", [("small", 0, False), ("demo", 0, False), ("small", 0, True), ("demo", 0, True),
                                                   ("procedural", 4000, False), ("procedural", 50000, False), ("procedural", 1 << 20, False)])
def test_device_tree_structure(dxrs, host, name, count, fast_build):
    """up to 4096 spheres: the host SAH topology with device-computed boxes (PT_FLAG_FAST_BUILD: the device LBVH); above: the
    device LBVH.  Either way the tree on the device is, record for record and box for box, the one the host builder makes."""
    t = dxrs.types
    kind = {"small": dxrs.host.SCENE_SMALL, "demo": dxrs.host.SCENE_DEMO, "procedural": dxrs.host.SCENE_PROCEDURAL}[name]
    spheres, materials, sd = host.scene(kind, seed=1, count=count)
    r = dxrs.Renderer(flags=t.PT_FLAG_FAST_BUILD if fast_build else 0)
    try:
        info = r.set_scene(spheres, materials, sd)
        sah = len(spheres) <= 4096 and not fast_build
        assert info.builder == (t.PT_BUILDER_HOST_SAH if sah else t.PT_BUILDER_DEVICE_LBVH)
        nodes, order = r.download_accel()
        if len(spheres) <= 100000:  # the Python invariant walk is O(n) with numpy per node: minutes at 2^20
            check_lbvh(spheres, nodes, order, info.depth)
        if name != "procedural":
            assert info.lds_resident == 1  # small scenes: whole BVH staged in LDS
        hn, ho, hd = dxrs.load_hip().lbvh_build_host(spheres, sah=sah)
        assert hd == info.depth and np.array_equal(ho, order)
        for f in ("child0", "child1", "parent", "lo0", "hi0", "lo1", "hi1"):
            assert np.array_equal(hn[f], nodes[f]), f  # device == host, boxes bit-for-bit
    finally:
        r.close()


def test_builders_give_identical_images(dxrs, host):
    """any valid BVH returns the same closest hit, so the SAH topology, the device LBVH and the host LBVH render the same bits
    (and count the same rays); a refit of the adopted SAH tree keeps that true after the spheres moved"""
    t = dxrs.types
    w, h = 320, 180
    cam, gs = host.camera(w, h), t.graphics_settings(w, h, bounces=6, spp=2)
    for kind, count in ((dxrs.host.SCENE_DEMO, 0), (dxrs.host.SCENE_PROCEDURAL, 3000)):
        spheres, materials, sd = host.scene(kind, seed=0, count=count)
        moved = spheres.copy(); moved["cy"] += 0.25 * np.sin(np.arange(len(spheres)))
        frames = []
        for flags in (0, t.PT_FLAG_FAST_BUILD, t.PT_FLAG_HOST_LBVH):
            r = dxrs.Renderer(flags=flags)
            try:
                info = r.set_scene(spheres, materials, sd)
                r.set_camera(cam); r.set_constants(gs)
                img, st = r.render()
                img2 = st2 = None
                if flags != t.PT_FLAG_HOST_LBVH:  # pt_update_spheres needs the device builder
                    r.update_spheres(moved)
                    img2, st2 = r.render()
                frames.append((info.builder, img, st.rays, img2, st2.rays if st2 else None))
            finally:
                r.close()
        assert [f[0] for f in frames] == [t.PT_BUILDER_HOST_SAH, t.PT_BUILDER_DEVICE_LBVH, t.PT_BUILDER_HOST_LBVH]
        for f in frames[1:]:
            assert f[2] == frames[0][2] and np.array_equal(f[1].view(np.uint32), frames[0][1].view(np.uint32))
        assert frames[1][4] == frames[0][4] and np.array_equal(frames[1][3].view(np.uint32), frames[0][3].view(np.uint32))
        assert not np.array_equal(frames[0][3], frames[0][1])  # the move is visible


def test_single_sphere_scene(dxrs, host, oracle):
    sph = np.zeros(1, dtype=dxrs.SPHERE_DTYPE); sph["r"] = 2.0; sph["cz"] = 1.0
    m = dxrs.types.default_material(1); m["BaseColor"] = (0.8, 0.3, 0.2, 1); m["Transmission"] = 1; m["Roughness"] = 0.1
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    gs = dxrs.types.graphics_settings(96, 64, bounces=6, spp=3)
    cam = host.camera(96, 64, position=(0, 0, -6))
    r = dxrs.Renderer()
    try:
        info = r.set_scene(sph, m, sd)
        assert info.node_count == 0 and info.depth == 0
        r.set_camera(cam); r.set_constants(gs)
        img, st = r.render()
        ref, ost = oracle.render(sph, m, sd, cam, gs, threads=4)
        assert st.rays == ost.rays and np.array_equal(img.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
    finally:
        r.close()


def test_million_sphere_frame_matches_oracle(dxrs, host, oracle, renderer):
    """BASELINE config C5's scene (2^20 procedural spheres + ground) at a small frame size: the whole pipeline (device LBVH
    build, global-memory traversal with ray replacement, split schedule, looping tail) against the oracle, which answers
    through its own, unrelated BVH -- bit-identical radiance and ray count"""
    spheres, materials, sd = host.scene(dxrs.host.SCENE_PROCEDURAL, seed=1, count=1 << 20)
    w, h = 192, 108
    cam = host.camera(w, h, jitter_index=3)
    for (bounces, spp) in ((8, 1), (3, 2)):
        gs = dxrs.types.graphics_settings(w, h, frame_index=3, bounces=bounces, spp=spp)
        renderer.set_scene(spheres, materials, sd); renderer.set_camera(cam); renderer.set_constants(gs)
        img, st = renderer.render()
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
        assert st.rays == ost.rays
        assert np.array_equal(img.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3])
