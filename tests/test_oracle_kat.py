"""Known-answer and property tests that pin the CPU oracle to analytic truth.
The reference has no tests or golden vectors for this path: these KATs are what the oracle is pinned by."""
import ctypes as C
import os
import math

import numpy as np
import pytest


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f32(x):
    return C.c_float(float(x))


def test_halton_known_values(oracle, host):
    lib = oracle.lib
    # HaltonSampler.ixx:32-34; Halton2D(1) = (1/2, 1/3), Halton3D(2) = (1/4, 2/3, 2/5)
    assert lib.oracle_halton(1, 2) == 0.5 and abs(lib.oracle_halton(1, 3) - 1 / 3) < 1e-7
    assert lib.oracle_halton(2, 2) == 0.25 and abs(lib.oracle_halton(2, 3) - 2 / 3) < 1e-7 and abs(lib.oracle_halton(2, 5) - 0.4) < 1e-7
    # radical inverse against an exact rational implementation
    from fractions import Fraction
    for base in (2, 3, 5):
        for i in range(1, 200):
            f, r, k = Fraction(1), Fraction(0), i
            while k:
                f /= base; r += f * (k % base); k //= base
            assert abs(lib.oracle_halton(i, base) - float(r)) < 2e-7
            assert lib.oracle_halton(i, base) == host.halton(i, base)  # host mirror == oracle


def test_frame0_jitter(host):
    cam = host.camera(1920, 1080, jitter_index=0)
    assert cam.Jitter[0] == 0.0 and abs(cam.Jitter[1] + 1 / 6) < 1e-7  # frame 0 -> (0, -1/6)


def test_rng_float_range_and_uniformity(oracle):
    lib = oracle.lib
    s = C.c_uint32(lib.oracle_rng_init(12, 34, 0))
    xs = np.array([lib.oracle_rng_float(C.byref(s)) for _ in range(200000)], dtype=np.float64)
    assert xs.min() > 0.0 and xs.max() <= 1.0
    hist, _ = np.histogram(xs, bins=64, range=(0, 1))
    chi2 = ((hist - len(xs) / 64) ** 2 / (len(xs) / 64)).sum()
    assert chi2 < 130  # 63 dof, p ~ 1e-6
    assert abs(xs.mean() - 0.5) < 5e-3
    # different pixels / frames decorrelate
    a = lib.oracle_rng_init(0, 0, 0); b = lib.oracle_rng_init(1, 0, 0); c = lib.oracle_rng_init(0, 1, 0); d = lib.oracle_rng_init(0, 0, 1)
    assert len({a, b, c, d}) == 4


def test_sincos_pow_accuracy(oracle):
    lib = oracle.lib
    s, c = C.c_float(), C.c_float()
    for u in np.linspace(0, 1, 4001):
        lib.oracle_sincos_2pi(f32(u), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(2 * math.pi * np.float32(u))) < 5e-7
        assert abs(c.value - math.cos(2 * math.pi * np.float32(u))) < 5e-7
    for u, es, ec in ((0.0, 0, 1), (0.25, 1, 0), (0.5, 0, -1), (0.75, -1, 0), (1.0, 0, 1)):
        lib.oracle_sincos_2pi(f32(u), C.byref(s), C.byref(c))
        assert s.value == es and c.value == ec
    for x in np.linspace(0.02, 1.0, 2000):
        got = lib.oracle_pow(f32(x), f32(2.4))
        assert abs(got - float(np.float32(x)) ** 2.4) <= 2e-6 * max(float(np.float32(x)) ** 2.4, 1e-3)
    assert lib.oracle_pow(f32(1.0), f32(2.4)) == 1.0
    for cc in (0.0, 0.02, 0.04045, 0.2, 0.5, 0.7, 1.0):
        exp = cc / 12.92 if cc <= 0.04045 else ((cc + 0.055) / 1.055) ** 2.4
        assert abs(lib.oracle_from_srgb(f32(cc)) - exp) < 2e-6


def test_sky_polynomial_matches_exact_formula(oracle, dxrs, host):
    """Spec S5: the procedural sky is a degree-7 polynomial fit of FromSrgb(lerp(1, (0.5, 0.7, 1), (d.y + 1) / 2))."""
    lib = oracle.lib
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    out = np.zeros(3, np.float32)
    for y in np.linspace(-1, 1, 2001):
        d = np.float32([math.sqrt(max(0.0, 1 - y * y)), y, 0])
        lib.oracle_sky(C.addressof(sd), fptr(d), fptr(out))
        t = (float(d[1]) + 1) / 2
        for ch, c in enumerate((0.5, 0.7, 1.0)):
            v = 1 + t * (c - 1)
            assert abs(out[ch] - ((v + 0.055) / 1.055) ** 2.4) < 1.5e-7
    assert out[2] == 1.0


def test_get_basis(oracle):
    lib = oracle.lib
    t, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
    n = np.float32([0, 0, 1])
    lib.oracle_get_basis(fptr(n), fptr(t), fptr(b))
    assert list(t) == [-1, 0, 0] and list(b) == [0, -1, 0]
    rng = np.random.default_rng(0)
    for _ in range(2000):
        n = rng.normal(size=3); n /= np.linalg.norm(n); n = n.astype(np.float32)
        lib.oracle_get_basis(fptr(n), fptr(t), fptr(b))
        m = np.stack([t, b, n]).astype(np.float64)
        assert np.allclose(m @ m.T, np.eye(3), atol=5e-6)
        assert np.linalg.det(m) > 0.99


def test_ray_sphere_hand_computed(oracle, dxrs):
    lib = oracle.lib
    sph = np.zeros(1, dtype=dxrs.SPHERE_DTYPE); sph["r"] = 1.0
    t = C.c_float()
    def hit(o, d, tmin=0.0, tmax=np.inf):
        o = np.float32(o); d = np.float32(d)
        return lib.oracle_intersect_sphere(fptr(o), fptr(d), f32(tmin), f32(tmax), sph.ctypes.data, C.byref(t)), t.value
    assert hit([0, 0, -3], [0, 0, 1]) == (1, 2.0)        # outside, head-on
    assert hit([0, 0, 0], [0, 0, 1]) == (1, 1.0)         # inside -> far root
    assert hit([0, 0, -3], [0, 0, -1])[0] == 0           # behind
    assert hit([0, 2, -3], [0, 0, 1])[0] == 0            # clear miss
    h, tv = hit([0, 1, -3], [0, 0, 1]); assert h == 1 and abs(tv - 3.0) < 1e-3  # tangent
    assert hit([0, 0, -3], [0, 0, 1], tmin=2.5) == (1, 4.0)  # near root below tmin -> far root
    assert hit([0, 0, -3], [0, 0, 1], tmax=1.5)[0] == 0
    # the demo's r = 50 ground under r = 0.075 spheres: hit point must lie on the sphere to ~1e-5
    sph["cx"], sph["cy"], sph["cz"], sph["r"] = 0, -50.1, 0, 50
    o = np.float32([3, 2, -15]); d = np.float32([0.1, -0.3, 1]); d /= np.linalg.norm(d)
    h, tv = hit(o, d); assert h == 1
    P, N, off, fr = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float(), C.c_int()
    lib.oracle_hit_frame(fptr(o), fptr(d), f32(tv), sph.ctypes.data, fptr(P), fptr(N), C.byref(off), C.byref(fr))
    assert abs(np.linalg.norm(P.astype(np.float64) - [0, -50.1, 0]) - 50) < 2e-5 and fr.value == 1
    assert 50 * 2 ** -16 * 0.99 < off.value < 60 * 2 ** -16
    # spawn on the correct side: a reflected ray does not re-hit, a transmitted one hits the far side
    up = np.float32(N); s_out = np.zeros(3, np.float32)
    lib.oracle_spawn_origin(fptr(P), fptr(N), off, fptr(up), fptr(s_out))
    assert lib.oracle_intersect_sphere(fptr(s_out), fptr(up), f32(0), f32(np.inf), sph.ctypes.data, C.byref(t)) == 0
    down = np.float32(-N)
    lib.oracle_spawn_origin(fptr(P), fptr(N), off, fptr(down), fptr(s_out))
    assert lib.oracle_intersect_sphere(fptr(s_out), fptr(down), f32(0), f32(np.inf), sph.ctypes.data, C.byref(t)) == 1 and abs(t.value - 100) < 1e-2


def test_fresnel_and_lobes(oracle, dxrs):
    lib = oracle.lib
    from oracle.binding import OracleBsdfOut
    for eta in (1.5, 1 / 1.5, 1.33):
        assert abs(lib.oracle_fresnel_dielectric(f32(eta), f32(1.0)) - ((eta - 1) / (eta + 1)) ** 2) < 1e-6  # normal incidence
    assert abs(lib.oracle_fresnel_dielectric(f32(1 / 1.5), f32(0.0)) - 1.0) < 1e-6                             # grazing (eta > 1 is caught by the TIR test first, BxDF.hlsli:155)
    m = dxrs.types.default_material(1)
    N = np.float32([0, 0, 1]); V = np.float32([0.3, 0.1, 0.9]); V /= np.linalg.norm(V)
    out = OracleBsdfOut()
    rng = np.random.default_rng(0)
    for metallic, transmission in ((0, 0), (1, 0), (0, 1), (0.5, 0.5), (0.2, 0.9)):
        m["BaseColor"] = (0.8, 0.6, 0.4, 1); m["Metallic"] = metallic; m["Transmission"] = transmission; m["Roughness"] = 0.4
        rnd = rng.random(4).astype(np.float32)
        lib.oracle_bsdf_step(m.ctypes.data, 1, fptr(N), fptr(V), fptr(rnd), C.byref(out))
        w = list(out.weights)
        assert abs(sum(w) - 1) < 1e-6 and all(x >= 0 for x in w)                    # BxDF.hlsli:188-195
        assert abs(w[2] - transmission * (1 - metallic)) < 1e-7
    # FindLobe boundaries (BxDF.hlsli:198-212): transmission first, then specular, else diffuse
    m["Metallic"] = 0; m["Transmission"] = 0.5
    lib.oracle_bsdf_step(m.ctypes.data, 1, fptr(N), fptr(V), fptr(np.float32([0.25, .5, .5, .5])), C.byref(out)); assert out.lobe == 2
    w = list(out.weights)
    lib.oracle_bsdf_step(m.ctypes.data, 1, fptr(N), fptr(V), fptr(np.float32([w[2] + 0.5 * w[1], .5, .5, .5])), C.byref(out)); assert out.lobe == 1
    lib.oracle_bsdf_step(m.ctypes.data, 1, fptr(N), fptr(V), fptr(np.float32([1.0, .5, .5, .5])), C.byref(out)); assert out.lobe == 0
    # TIR switch (BxDF.hlsli:155): from inside glass at a grazing angle the transmission lobe reflects
    m["Transmission"] = 1; m["Roughness"] = 0
    Vg = np.float32([0.95, 0, 0.3122499]); Vg /= np.linalg.norm(Vg)
    lib.oracle_bsdf_step(m.ctypes.data, 0, fptr(np.float32([0, 0, -1])), fptr(Vg), fptr(np.float32([0.1, .3, .6, .999])), C.byref(out))
    assert out.lobe == 2 and out.L[2] > 0  # stayed on V's side: reflected


@pytest.mark.parametrize("roughness", [0.1, 0.5, 1.0])
def test_pdfs_integrate_to_one(oracle, roughness):
    """int pdf dw = 1 per lobe.  Cosine lobe: NoL/pi over the hemisphere.  VNDF reflection lobe: integrate over the
    half-vector domain, pdf_L dw_L = pdf_L * 4 (V.H) dw_H, on a grid concentrated around the GGX peak."""
    lib = oracle.lib
    rng = np.random.default_rng(1)
    z = rng.random(200000)
    assert abs((z / np.pi).mean() * 2 * np.pi - 1) < 5e-3  # uniform hemisphere sampler, pdf 1/2pi
    V = np.float32([0.5, 0.2, 0.8]); V /= np.linalg.norm(V)
    nt, nphi = 1500, 96
    edges = (np.arange(nt + 1) / nt) ** 3 * (np.pi / 2)
    theta = 0.5 * (edges[1:] + edges[:-1]); dtheta = np.diff(edges)
    phi = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
    total = 0.0
    for ti, dt in zip(theta, dtheta):
        st, ct = np.sin(ti), np.cos(ti)
        pdf = lib.oracle_vndf_pdf(fptr(V), f32(ct), f32(roughness))  # depends on H only through NoH
        voh = st * np.cos(phi) * V[0] + st * np.sin(phi) * V[1] + ct * V[2]
        total += pdf * 4.0 * np.clip(voh, 0, None).sum() * st * dt * (2 * np.pi / nphi)
    assert abs(total - 1.0) < 2e-2, total


def test_white_furnace(oracle, dxrs, host):
    """White furnace on the specular lobe: BaseColor 1, Metallic 1 (F = 1) sphere in a constant white environment.
    Per bounce the weight is f/pdf = G2/G1 <= 1 (height-correlated Smith over VNDF sampling), so radiance <= env and it
    approaches env as the roughness goes to 0 (RR and the throughput cutoff off).  (The diffuse lobe is Burley's, which is
    not energy conserving by design, so it has no such bound.)"""
    sph = np.zeros(1, dtype=dxrs.SPHERE_DTYPE); sph["cz"] = 0; sph["r"] = 1
    m = dxrs.types.default_material(1); m["BaseColor"] = (1, 1, 1, 1); m["Metallic"] = 1; m["Roughness"] = 0.6
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    sd.EnvironmentLightColor[0] = sd.EnvironmentLightColor[1] = sd.EnvironmentLightColor[2] = 1.0; sd.EnvironmentLightColor[3] = 1.0
    cam = host.camera(32, 32, position=(0, 0, -3), jitter=False)
    means = {}
    for roughness in (0.6, 0.05):
        m["Roughness"] = roughness
        for bounces in (1, 8):
            gs = dxrs.types.graphics_settings(32, 32, bounces=bounces, spp=64, rr=False, threshold=0.0)
            img, _ = oracle.render(sph, m, sd, cam, gs, rect=(12, 12, 8, 8), threads=4)  # pixels on the sphere
            assert img[..., :3].max() <= 1.0 + 1e-4
            means[(roughness, bounces)] = float(img[..., :3].mean())
    # a convex sphere has no inter-reflection: every reflected ray reaches the environment on bounce 1
    assert means[(0.6, 1)] == means[(0.6, 8)] and means[(0.05, 1)] == means[(0.05, 8)]
    assert 0.6 < means[(0.6, 1)] < 1.0 and means[(0.05, 1)] > 0.97, means


def test_rng_golden_vectors(oracle):
    """First 16 uint outputs for 8 (pixel, frame) seeds: committed golden (tests/golden/rng_streams.npy)."""
    import os
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "rng_streams.npy"))
    seeds = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1919, 1079, 0), (3839, 2159, 7), (65535, 65535, 0xFFFFFFFF), (123, 456, 789)]
    lib = oracle.lib
    got = np.zeros((8, 17), dtype=np.uint32)
    for i, (x, y, f) in enumerate(seeds):
        s = C.c_uint32(lib.oracle_rng_init(x, y, f)); got[i, 0] = s.value
        for k in range(16):
            got[i, 1 + k] = lib.oracle_rng_next(C.byref(s))
    assert np.array_equal(got, golden)
    # independent pure-Python restatement of SURVEY Appendix A
    def h(x):
        x &= 0xFFFFFFFF; x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16; return x
    for i, (x, y, f) in enumerate(seeds):
        seed = h((f + 0x035F9F29) & 0xFFFFFFFF); v = ((x << 16) | y) & 0xFFFFFFFF
        st = seed ^ ((h(v) + 0x9E3779B9 + ((seed << 6) & 0xFFFFFFFF) + (seed >> 2)) & 0xFFFFFFFF)
        assert st == int(golden[i, 0])
        for k in range(16):
            st = h(st); assert st == int(golden[i, 1 + k])


def test_oracle_bvh_equals_brute_force(dxrs, host, oracle):
    """the oracle's own median-split BVH (used above 64 spheres; it makes the 2^20-sphere configuration runnable on the CPU) returns
    exactly the brute-force closest hit -- t bit for bit and the lowest id on ties -- on overlapping, nested, coincident and
    wildly different-sized spheres, for rays from outside, from inside, from sphere surfaces and parallel to the axes"""
    rng = np.random.default_rng(42)
    for style in range(4):
        n = [300, 1000, 80, 5000][style]
        s = np.zeros(n, dtype=dxrs.SPHERE_DTYPE)
        if style == 0:
            s["cx"], s["cy"], s["cz"] = rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-8, 8, n)
            s["r"] = np.exp(rng.uniform(np.log(0.05), np.log(3.0), n))
        elif style == 1:
            s["cx"], s["cy"], s["cz"] = rng.uniform(-50, 50, n), rng.uniform(0, 2, n), rng.uniform(-50, 50, n)
            s["r"] = rng.uniform(0.02, 0.3, n)
            s[0] = (0, -1000.2, 0, 1000.0)
        elif style == 2:  # coincident duplicates: ties must go to the lowest id
            base = 20
            s["cx"][:base], s["cy"][:base], s["cz"][:base] = rng.uniform(-4, 4, base), rng.uniform(-3, 3, base), rng.uniform(-6, 6, base)
            s["r"][:base] = rng.uniform(0.3, 2.0, base)
            for i in range(base, n):
                s[i] = s[rng.integers(0, base)]
        else:
            s["cx"], s["cy"], s["cz"] = rng.normal(0, 3, n), rng.normal(0, 3, n), rng.normal(0, 3, n)
            s["r"] = np.exp(rng.uniform(np.log(1e-3), np.log(1.0), n))
        m = 1500
        c = np.stack([s["cx"], s["cy"], s["cz"]], 1).astype(np.float64)
        o = rng.uniform(c.min(0) - 2, c.max(0) + 2, (m, 3))
        pick = rng.integers(0, n, m)
        d = np.where((np.arange(m) % 2 == 0)[:, None], c[pick] + rng.normal(size=(m, 3)) * s["r"][pick, None] * 0.7 - o, rng.normal(size=(m, 3)))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        k = np.arange(m) % 4 == 1  # origins on sphere surfaces (the secondary-ray case)
        nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        o[k] = c[pick[k]] + nrm[k] * s["r"][pick[k], None] * (1 + 2 ** -14)
        d32 = d.astype(np.float32)
        ax = np.arange(m) % 16 == 3
        d32[ax, rng.integers(0, 3)] = 0.0
        d32 /= np.linalg.norm(d32.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
        tb, ib = oracle.closest_hits(s, o, d32, use_bvh=False)
        ta, ia = oracle.closest_hits(s, o, d32, use_bvh=True)
        assert np.array_equal(ib, ia) and np.array_equal(tb.view(np.uint32), ta.view(np.uint32)), style
        assert (ib != 0xFFFFFFFF).mean() > 0.2


def test_oracle_frame_same_with_and_without_bvh(dxrs, host, oracle, monkeypatch):
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)  # 441 spheres: above the BVH threshold
    w, h = 96, 54
    cam, gs = host.camera(w, h), dxrs.types.graphics_settings(w, h, bounces=5, spp=2)
    a, sa = oracle.render(spheres, materials, sd, cam, gs, threads=8)
    monkeypatch.setenv("ORACLE_NO_BVH", "1")
    b, sb = oracle.render(spheres, materials, sd, cam, gs, threads=8)
    assert sa.rays == sb.rays and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_reproduces_the_golden_crops(dxrs, host, oracle):
    """the committed rendered fixtures (tests/golden_cases.py: C1, C2, textured + environment map, direct illumination, cube
    environment, and the tone-mapped C2 crop) pin the oracle: any change of its arithmetic shows up here, on the CPU"""
    import golden_cases
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    for c in golden_cases.cases(dxrs, host):
        img, _ = oracle.render(c["spheres"], c["materials"], c["sd"], c["cam"], c["gs"], rect=c["rect"], threads=8, textures=c["textures"])
        gold = np.load(os.path.join(gold_dir, c["file"]))
        assert np.array_equal(img.view(np.uint32), gold.view(np.uint32)), c["file"]
        assert np.isfinite(gold).all() and gold[..., :3].std() > 0.01, c["file"]  # a crop with content
    src, dst, params = golden_cases.tonemap_case(dxrs)
    assert np.array_equal(oracle.tonemap(np.load(os.path.join(gold_dir, src)), params), np.load(os.path.join(gold_dir, dst)))


def test_empty_scene_is_the_environment(dxrs, host, oracle):
    """A scene without spheres (a TLAS without instances) is legal: every ray misses and every pixel is the environment, exactly the
    frame of a scene whose only sphere lies behind the camera; one primary ray per pixel is counted, whatever the sample count."""
    s, m, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    w, h = 64, 40
    for spp, bounces in ((1, 8), (3, 2)):
        gs = dxrs.types.graphics_settings(w, h, frame_index=2, bounces=bounces, spp=spp)
        cam = host.camera(w, h, jitter_index=1)
        img, st = oracle.render(s[:0], m[:0], sd, cam, gs, threads=2)
        behind = s[:1].copy()
        behind["cx"], behind["cy"], behind["cz"], behind["r"] = 0.0, 0.0, -1e4, 0.5
        ref, st_ref = oracle.render(behind, m[:1], sd, cam, gs, threads=2)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.rays == st_ref.rays == w * h
        assert np.all(img[..., 3] == 1.0) and np.isfinite(img).all() and img[..., :3].min() > 0.0
