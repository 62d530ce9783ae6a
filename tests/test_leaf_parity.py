"""CPU-side bit-parity of the product's device leaf arithmetic (csrc/pt_math.h, csrc/pt_bsdf.h compiled as host C++
by tests/hostshim) against the CPU oracle (oracle/pt_oracle.c), function by function, on seeded random inputs.
Bar: bit-exact (the two are independent restatements of DESIGN.md's frozen arithmetic spec; a one-ulp difference in
any of these flips branch decisions of the estimator on the GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.binding import OracleBsdfOut, declare_leaf_api

HERE = os.path.dirname(os.path.abspath(__file__))
N = 20000


@pytest.fixture(scope="module")
def dev():
    lib = C.CDLL(os.path.join(HERE, "hostshim", "libdevmath_host.so"))
    declare_leaf_api(lib, "dev_")
    return lib


@pytest.fixture(scope="module")
def ora(oracle):
    return oracle.lib


def f32(x):
    return C.c_float(float(x))


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


def test_rng_stream_bit_exact(dev, ora):
    rng = np.random.default_rng(1)
    for _ in range(200):
        px, py, frame = int(rng.integers(0, 65536)), int(rng.integers(0, 65536)), int(rng.integers(0, 2**32))
        a, b = C.c_uint32(ora.oracle_rng_init(px, py, frame)), C.c_uint32(dev.dev_rng_init(px, py, frame))
        assert a.value == b.value
        for _ in range(16):
            assert ora.oracle_rng_next(C.byref(a)) == dev.dev_rng_next(C.byref(b))
        for _ in range(8):
            fa, fb = ora.oracle_rng_float(C.byref(a)), dev.dev_rng_float(C.byref(b))
            assert fa == fb and 0.0 < fa <= 1.0


def test_scalar_functions_bit_exact(dev, ora):
    rng = np.random.default_rng(2)
    u = np.concatenate([rng.random(N).astype(np.float32), np.float32([0.0, 1.0, 0.25, 0.5, 0.75, 0.125, 1e-7, 0.99999994])])
    sa, ca, sb, cb = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for x in u:
        ora.oracle_sincos_2pi(f32(x), C.byref(sa), C.byref(ca))
        dev.dev_sincos_2pi(f32(x), C.byref(sb), C.byref(cb))
        assert same_bits([sa.value, ca.value], [sb.value, cb.value])
    x = np.concatenate([np.exp(rng.uniform(-20, 20, N)).astype(np.float32), np.float32([1.0, 0.5, 2.0, 0.526, 1.4142135, 1.4142137])])
    for v in x:
        assert same_bits(ora.oracle_log2(f32(v)), dev.dev_log2(f32(v)))
    y = np.concatenate([rng.uniform(-30, 30, N).astype(np.float32), np.float32([0.0, -0.5, 0.5, -2.4, 1.0])])
    for v in y:
        assert same_bits(ora.oracle_exp2(f32(v)), dev.dev_exp2(f32(v)))
    c = np.concatenate([rng.uniform(-0.2, 1.2, N).astype(np.float32), np.float32([0.0, 1.0, 0.04045, 0.5, 0.7])])
    for v in c:
        assert same_bits(ora.oracle_from_srgb(f32(v)), dev.dev_from_srgb(f32(v)))


def test_brdf_terms_bit_exact(dev, ora):
    rng = np.random.default_rng(3)
    rough = np.concatenate([rng.uniform(2e-3, 1.0, N // 2), np.full(N // 2, 2e-3)]).astype(np.float32)
    a = rng.random(N).astype(np.float32)
    b = rng.random(N).astype(np.float32)
    c = rng.random(N).astype(np.float32)
    a[:50] = 1.0  # NoH == 1 exactly: the case that breaks the textbook GGX form (DESIGN.md deviation D1)
    eta = rng.choice(np.float32([1.5, 1 / 1.5, 1.33, 1 / 1.33, 1.0]), N)
    for i in range(N):
        r = f32(rough[i])
        assert same_bits(ora.oracle_distribution_term(r, f32(a[i])), dev.dev_distribution_term(r, f32(a[i])))
        assert same_bits(ora.oracle_geometry_term_mod(r, f32(a[i]), f32(b[i])), dev.dev_geometry_term_mod(r, f32(a[i]), f32(b[i])))
        assert same_bits(ora.oracle_diffuse_term(r, f32(a[i]), f32(b[i]), f32(c[i])), dev.dev_diffuse_term(r, f32(a[i]), f32(b[i]), f32(c[i])))
        assert same_bits(ora.oracle_fresnel_dielectric(f32(eta[i]), f32(a[i])), dev.dev_fresnel_dielectric(f32(eta[i]), f32(a[i])))
    d = float(ora.oracle_distribution_term(f32(2e-3), f32(1.0)))
    assert np.isfinite(d) and d > 0


def test_vector_functions_bit_exact(dev, ora):
    rng = np.random.default_rng(4)
    n = 5000
    nrm = unit(rng, n)
    nrm[0] = (0, 0, 1); nrm[1] = (0, 0, -1); nrm[2] = (1, 0, 0)
    u2 = rng.random((n, 2)).astype(np.float32)
    rough = rng.uniform(2e-3, 1.0, n).astype(np.float32)
    vl = unit(rng, n); vl[:, 2] = np.abs(vl[:, 2])
    f0 = rng.random((n, 3)).astype(np.float32)
    o3 = lambda: np.zeros(3, dtype=np.float32)
    for i in range(n):
        t1, b1, t2, b2 = o3(), o3(), o3(), o3()
        ora.oracle_get_basis(fptr(nrm[i]), fptr(t1), fptr(b1)); dev.dev_get_basis(fptr(nrm[i]), fptr(t2), fptr(b2))
        assert same_bits(t1, t2) and same_bits(b1, b2)
        r1, r2 = o3(), o3()
        ora.oracle_cosine_ray(fptr(u2[i]), fptr(r1)); dev.dev_cosine_ray(fptr(u2[i]), fptr(r2))
        assert same_bits(r1, r2)
        ora.oracle_vndf_ray(fptr(u2[i]), f32(rough[i]), fptr(vl[i]), fptr(r1)); dev.dev_vndf_ray(fptr(u2[i]), f32(rough[i]), fptr(vl[i]), fptr(r2))
        assert same_bits(r1, r2)
        assert same_bits(ora.oracle_vndf_pdf(fptr(vl[i]), f32(u2[i, 0]), f32(rough[i])), dev.dev_vndf_pdf(fptr(vl[i]), f32(u2[i, 0]), f32(rough[i])))
        ora.oracle_environment_term_rtg(fptr(f0[i]), f32(u2[i, 1]), f32(rough[i]), fptr(r1)); dev.dev_environment_term_rtg(fptr(f0[i]), f32(u2[i, 1]), f32(rough[i]), fptr(r2))
        assert same_bits(r1, r2)


def test_sky_and_primary_ray_bit_exact(dev, ora, dxrs, host):
    rng = np.random.default_rng(5)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    d = unit(rng, 3000)
    a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for i in range(len(d)):
        ora.oracle_sky(C.addressof(sd), fptr(d[i]), fptr(a)); dev.dev_sky(C.addressof(sd), fptr(d[i]), fptr(b))
        assert same_bits(a, b)
    sd.EnvironmentLightColor[0], sd.EnvironmentLightColor[1], sd.EnvironmentLightColor[2], sd.EnvironmentLightColor[3] = 0.25, 0.5, 0.75, 1.0
    ora.oracle_sky(C.addressof(sd), fptr(d[0]), fptr(a)); dev.dev_sky(C.addressof(sd), fptr(d[0]), fptr(b))
    assert list(a) == [0.25, 0.5, 0.75] and same_bits(a, b)
    for (w, h, ji) in ((1920, 1080, 0), (3840, 2160, 3), (256, 256, 7), (97, 53, 1)):
        cam = host.camera(w, h, jitter_index=ji)
        for _ in range(500):
            px, py = int(rng.integers(0, w)), int(rng.integers(0, h))
            o1, d1, o2, d2 = (np.zeros(3, np.float32) for _ in range(4))
            t0a, t1a, t0b, t1b = C.c_float(), C.c_float(), C.c_float(), C.c_float()
            ora.oracle_primary_ray(C.addressof(cam), px, py, w, h, fptr(o1), fptr(d1), C.byref(t0a), C.byref(t1a))
            dev.dev_primary_ray(C.addressof(cam), px, py, w, h, fptr(o2), fptr(d2), C.byref(t0b), C.byref(t1b))
            assert same_bits(o1, o2) and same_bits(d1, d2) and same_bits(t0a.value, t0b.value) and t1a.value == t1b.value == np.inf


def test_sphere_intersection_bit_exact(dev, ora, dxrs):
    rng = np.random.default_rng(6)
    n = 20000
    sph = np.zeros(n, dtype=dxrs.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-10, 10, n)
    sph["r"] = np.exp(rng.uniform(np.log(0.02), np.log(50), n))
    o = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    target = np.stack([sph["cx"], sph["cy"], sph["cz"]], 1) + rng.normal(size=(n, 3)) * sph["r"][:, None] * 0.8
    d = (target - o); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
    hits = 0
    for i in range(n):
        s = sph[i:i + 1]
        ta, tb = C.c_float(-1), C.c_float(-1)
        ha = ora.oracle_intersect_sphere(fptr(o[i]), fptr(d[i]), f32(0), f32(np.inf), s.ctypes.data, C.byref(ta))
        hb = dev.dev_intersect_sphere(fptr(o[i]), fptr(d[i]), f32(0), f32(np.inf), s.ctypes.data, C.byref(tb))
        assert ha == hb and same_bits(ta.value, tb.value)
        if ha:
            hits += 1
            P1, N1, P2, N2 = (np.zeros(3, np.float32) for _ in range(4))
            of1, of2, fr1, fr2 = C.c_float(), C.c_float(), C.c_int(), C.c_int()
            ora.oracle_hit_frame(fptr(o[i]), fptr(d[i]), ta, s.ctypes.data, fptr(P1), fptr(N1), C.byref(of1), C.byref(fr1))
            dev.dev_hit_frame(fptr(o[i]), fptr(d[i]), tb, s.ctypes.data, fptr(P2), fptr(N2), C.byref(of2), C.byref(fr2))
            assert same_bits(P1, P2) and same_bits(N1, N2) and same_bits(of1.value, of2.value) and fr1.value == fr2.value
            L = unit(rng, 1)[0]
            s1, s2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
            ora.oracle_spawn_origin(fptr(P1), fptr(N1), of1, fptr(L), fptr(s1)); dev.dev_spawn_origin(fptr(P2), fptr(N2), of2, fptr(L), fptr(s2))
            assert same_bits(s1, s2)
    assert hits > n // 4


def test_bsdf_step_bit_exact(dev, ora, dxrs):
    """Whole BSDF interaction (lobe weights, lobe choice, sampled direction, pdf, value) for random materials."""
    rng = np.random.default_rng(7)
    n = 20000
    mats = dxrs.types.default_material(n)
    mats["BaseColor"][:, :3] = rng.random((n, 3))
    mats["Metallic"] = rng.choice([0.0, 1.0, 0.5], n) * rng.random(n) ** 0.3
    mats["Roughness"] = rng.choice([0.0, 1.0], n) * rng.random(n)
    mats["Transmission"] = rng.choice([0.0, 1.0, 0.5], n)
    mats["IOR"] = rng.choice([1.5, 1.33, 1.0, 2.4], n)
    Ng = unit(rng, n)
    V = unit(rng, n)
    flip = (np.einsum("ij,ij->i", Ng, V) < 0)
    front = rng.integers(0, 2, n)
    # V must be on the side of the shading normal: shading normal = front ? Ng : -Ng
    want_pos = front == 1
    V[(flip & want_pos) | (~flip & ~want_pos)] *= -1
    rnd = rng.random((n, 4)).astype(np.float32)
    lobes = np.zeros(3, dtype=int)
    for i in range(n):
        a, b = OracleBsdfOut(), OracleBsdfOut()
        m = mats[i:i + 1]
        ora.oracle_bsdf_step(m.ctypes.data, int(front[i]), fptr(Ng[i]), fptr(V[i]), fptr(rnd[i]), C.byref(a))
        dev.dev_bsdf_step(m.ctypes.data, int(front[i]), fptr(Ng[i]), fptr(V[i]), fptr(rnd[i]), C.byref(b))
        assert a.lobe == b.lobe and a.valid == b.valid
        assert same_bits(list(a.weights), list(b.weights))
        assert same_bits(list(a.L), list(b.L)) and same_bits(a.pdf, b.pdf) and same_bits(list(a.f), list(b.f))
        lobes[a.lobe] += 1
    assert (lobes > n // 20).all()  # all three lobes exercised
