"""Row a18's texture branch -- GetEnvironmentLightColor with a lat-long environment map (Shaders/ShadingHelpers.hlsli:13-24,
Math::ToLatLongCoordinate Math.hlsli:29-33): SceneData.EnvironmentLightTextureDescriptor indexes the texture table of
pt_set_textures, the lookup direction is rotated by the upper 3x3 of EnvironmentLightTransform and normalised, the map is
sampled at level 0 (bilinear, wrap).  CPU: the coordinate convention, the rotation semantics and the oracle's consistency
with the constant-colour branch.  GPU: whole frames through the C-ABI against the oracle, bit-exact."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from oracle.binding import declare_leaf_api
from test_textures import bits, call3, fa, make_textured_scene, unit
from util import count_mismatch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    lib = C.CDLL(os.path.join(HERE, "hostshim", "libdevmath_host.so"))
    declare_leaf_api(lib, "dev_")
    return lib


def set_env(sd, descriptor, matrix=None):
    """a copy of sd with the environment texture descriptor (and optionally the 3x3 of EnvironmentLightTransform) set"""
    out = copy.copy(sd)
    out.EnvironmentLightTextureDescriptor = descriptor
    m = np.eye(3) if matrix is None else np.asarray(matrix, dtype=np.float64)
    for r in range(3):
        for k in range(3):
            out.EnvironmentLightTransform[4 * r + k] = float(m[r, k])
        out.EnvironmentLightTransform[4 * r + 3] = 123.0  # the translation column is ignored ((float3x3) cast)
    return out


def rot_y(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def random_rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q


def test_latlong_convention_and_parity(oracle, dev):
    uv = lambda d: call3(oracle.lib.oracle_latlong_uv, fa(*d), n_out=2)
    assert np.allclose(uv((0, 0, 1)), (0.5, 0.5), atol=1e-6)     # +z: centre of the map
    assert np.allclose(uv((1, 0, 0)), (0.75, 0.5), atol=1e-5)    # +x: a quarter turn to the right
    assert np.allclose(uv((-1, 0, 0)), (0.25, 0.5), atol=1e-5)
    assert np.allclose(uv((0, 0, -1)), (1.0, 0.5), atol=1e-6)    # the seam (wrap addressing joins u = 1 and u = 0)
    assert np.allclose(uv((0, 1, 0))[1], 0.0, atol=1e-6) and np.allclose(uv((0, -1, 0))[1], 1.0, atol=1e-6)
    rng = np.random.default_rng(11)
    dirs = np.concatenate([unit(rng, 4000), [[0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0], [1e-20, 1, 0]]]).astype(np.float32)
    for d in dirs:
        a, b = uv(d), call3(dev.dev_latlong_uv, fa(*d), n_out=2)
        assert np.array_equal(bits(a), bits(b))
        dd = d.astype(np.float64)
        if abs(dd[1]) < 0.9999:
            want_u = (1 + np.arctan2(dd[0], dd[2]) / np.pi) / 2
            assert min(abs(a[0] - want_u), 1 - abs(a[0] - want_u)) < 1e-5
        assert abs(a[1] - np.arccos(np.clip(dd[1], -1, 1)) / np.pi) < 2e-4  # acos through atan2(sqrt(1 - y^2), y): sqrt(eps) near the poles


def small_scene(dxrs, host):
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    return spheres, materials, sd


def test_oracle_constant_map_equals_constant_colour(dxrs, host, oracle):
    """a one-colour HDR map under any transform == EnvironmentLightColor = that colour (bilinear of equal texels is exact)"""
    from dxrs_amd import textures as T
    spheres, materials, sd = small_scene(dxrs, host)
    w, h = 64, 48
    cam, gs = host.camera(w, h), dxrs.types.graphics_settings(w, h, bounces=4, spp=2)
    colour = np.array([1.75, 0.5, 3.25], np.float32)
    ts = T.TextureSet(len(spheres))
    idx = ts.add_hdr_image(np.tile(colour, (4, 8, 1)))
    const = copy.copy(sd)
    for k in range(3):
        const.EnvironmentLightColor[k] = float(colour[k])
    const.EnvironmentLightColor[3] = 1.0
    ref, st = oracle.render(spheres, materials, const, cam, gs, threads=4)
    rng = np.random.default_rng(3)
    for m in (None, random_rotation(rng), 3.0 * random_rotation(rng)):
        img, st2 = oracle.render(spheres, materials, set_env(sd, idx, m), cam, gs, threads=4, textures=ts)
        assert st.rays == st2.rays and np.array_equal(bits(img), bits(ref))


def test_oracle_rotation_about_y_shifts_the_map(dxrs, host, oracle):
    """RotateVector = mul(M, v): a turn by alpha about +y moves the lookup by alpha / 2pi in u, i.e. the frame equals the one
    of the identity transform with the map rolled by alpha / 2pi * width texels (up to fp32 noise in the direction)"""
    from dxrs_amd import textures as T
    spheres, materials, sd = small_scene(dxrs, host)
    w, h = 96, 64
    cam, gs = host.camera(w, h), dxrs.types.graphics_settings(w, h, bounces=0)  # primary hits + misses only: no sampling decisions
    env = T.sky_latlong(64, 32, seed=5, sun_radiance=0.0)
    shift = 9
    ts_a, ts_b = T.TextureSet(len(spheres)), T.TextureSet(len(spheres))
    ia = ts_a.add_hdr_image(env)
    ib = ts_b.add_hdr_image(np.roll(env, -shift, axis=1))  # rolled[x] = env[x + shift]
    a, _ = oracle.render(spheres, materials, set_env(sd, ia, rot_y(2 * np.pi * shift / 64)), cam, gs, threads=4, textures=ts_a)
    b, _ = oracle.render(spheres, materials, set_env(sd, ib), cam, gs, threads=4, textures=ts_b)
    assert np.allclose(a, b, rtol=0, atol=2e-3 * float(env.max()))
    plain, _ = oracle.render(spheres, materials, set_env(sd, ia), cam, gs, threads=4, textures=ts_a)
    assert np.abs(plain - a).max() > 0.05  # the rotation does something


def test_oracle_primary_miss_samples_the_map(dxrs, host, oracle):
    """camera at the origin looking down +z with nothing in front: the centre pixel reads the map's centre, the image's left
    edge reads texels left of it (u < 0.5), its top reads v < 0.5"""
    from dxrs_amd import textures as T
    t = dxrs.types
    s = np.zeros(1, dtype=dxrs.SPHERE_DTYPE); s[0] = (0, 0, -50.0, 1.0)  # behind the camera
    m = t.default_material(1)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    W, H = 256, 128
    u = (np.arange(W) + 0.5) / W; v = (np.arange(H) + 0.5) / H
    env = np.zeros((H, W, 3), np.float32)
    env[..., 0] = u[None, :]; env[..., 1] = v[:, None]; env[..., 2] = 2.0  # the map stores its own coordinates
    ts = T.TextureSet(1); idx = ts.add_hdr_image(env)
    w, h = 65, 33
    cam = host.camera(w, h, position=(0.0, 0.0, 0.0), jitter=False)
    gs = t.graphics_settings(w, h, bounces=2)
    img, _ = oracle.render(s, m, set_env(sd, idx), cam, gs, threads=2, textures=ts)
    assert np.all(img[..., 2] == 2.0) and np.all(img[..., 3] == 1.0)
    cu, cv = img[h // 2, w // 2, 0], img[h // 2, w // 2, 1]
    assert abs(cu - 0.5) < 0.02 and abs(cv - 0.5) < 0.02
    assert np.all(np.diff(img[h // 2, :, 0]) > 0)   # u grows to the right (towards +x)
    assert np.all(np.diff(img[:, w // 2, 1]) > 0)   # v grows downwards (towards -y)


def test_oracle_rejects_bad_environment_descriptors(dxrs, host, oracle):
    from dxrs_amd import textures as T
    spheres, materials, sd = small_scene(dxrs, host)
    cam, gs = host.camera(32, 32), dxrs.types.graphics_settings(32, 32, bounces=1)
    ts = T.TextureSet(len(spheres)); ts.add_hdr_image(np.ones((2, 2, 3), np.float32))
    with pytest.raises(Exception):
        oracle.render(spheres, materials, set_env(sd, 0), cam, gs)              # no table
    with pytest.raises(Exception):
        oracle.render(spheres, materials, set_env(sd, 1), cam, gs, textures=ts)  # out of range
    cube = set_env(sd, 0); cube.IsEnvironmentLightTextureCubeMap = 1
    with pytest.raises(Exception):
        oracle.render(spheres, materials, cube, cam, gs, textures=ts)  # a cube map needs six table entries
    for k in range(4):
        ts.add_hdr_image(np.ones((2, 2, 3), np.float32))
    ts.add_hdr_image(np.ones((2, 3, 3), np.float32))
    with pytest.raises(Exception):
        oracle.render(spheres, materials, cube, cam, gs, textures=ts)  # ... square and of one size


# ---- cube maps ---------------------------------------------------------------------------------------------------------

def set_cube(sd, descriptor, matrix=None):
    out = set_env(sd, descriptor, matrix)
    out.IsEnvironmentLightTextureCubeMap = 1
    return out


def test_cube_face_convention_and_parity(oracle, dev):
    def face_uv(fn, d):
        uv = (C.c_float * 2)()
        return int(fn(fa(*d), uv)), np.array(uv[:], dtype=np.float32)
    o = lambda d: face_uv(oracle.lib.oracle_cube_face_uv, d)
    # the axes hit the centres of the faces, in D3D order +X -X +Y -Y +Z -Z
    for face, axis in enumerate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]):
        f, uv = o(axis)
        assert f == face and np.allclose(uv, 0.5)
    # D3D face coordinates: on +X u runs towards -z and v towards -y; on +Y u towards +x, v towards +z; on +Z u towards +x, v towards -y
    assert np.allclose(o((1, 0.5, 0))[1], (0.5, 0.25)) and np.allclose(o((1, 0, 0.5))[1], (0.25, 0.5))
    assert np.allclose(o((-1, 0, 0.5))[1], (0.75, 0.5))
    assert np.allclose(o((0.5, 1, 0))[1], (0.75, 0.5)) and np.allclose(o((0, 1, 0.5))[1], (0.5, 0.75))
    assert np.allclose(o((0, -1, 0.5))[1], (0.5, 0.25))
    assert np.allclose(o((0.5, 0, 1))[1], (0.75, 0.5)) and np.allclose(o((0, 0.5, 1))[1], (0.5, 0.25))
    assert np.allclose(o((0.5, 0, -1))[1], (0.25, 0.5))
    # ties: z over y over x
    assert o((1, 1, 1))[0] == 4 and o((1, 1, -1))[0] == 5 and o((1, 1, 0.5))[0] == 2 and o((1, -1, 0))[0] == 3
    rng = np.random.default_rng(12)
    dirs = np.concatenate([unit(rng, 4000), [[1, 1, 1], [-1, 1, -1], [0, 0, 0], [1e-30, 0, 0], [1, -1, 0.999999]]]).astype(np.float32)
    from dxrs_amd.textures import cube_directions
    for d in dirs:
        (f1, uv1), (f2, uv2) = o(d), face_uv(dev.dev_cube_face_uv, d)
        assert f1 == f2 and np.array_equal(bits(uv1), bits(uv2))
    # cube_directions is the inverse map: the direction through texel (face, y, x) lands on that texel's centre
    size = 4
    cd = cube_directions(size)
    for face in range(6):
        for y in range(size):
            for x in range(size):
                f, uv = o(cd[face, y, x].astype(np.float32))
                assert f == face and np.allclose(uv, ((x + 0.5) / size, (y + 0.5) / size), atol=1e-6)


def test_oracle_cube_map_consistency(dxrs, host, oracle):
    """a one-colour cube == the constant-colour branch, bit for bit; a cube and a lat-long map made from the same smooth
    function of the direction light the scene alike"""
    from dxrs_amd import textures as T
    spheres, materials, sd = small_scene(dxrs, host)
    w, h = 64, 48
    cam, gs = host.camera(w, h), dxrs.types.graphics_settings(w, h, bounces=0)
    colour = np.array([0.5, 2.25, 1.0], np.float32)
    ts = T.TextureSet(len(spheres))
    first = ts.add_cube([np.tile(colour, (3, 3, 1))] * 6)
    const = copy.copy(sd)
    for k in range(3):
        const.EnvironmentLightColor[k] = float(colour[k])
    const.EnvironmentLightColor[3] = 1.0
    gs4 = dxrs.types.graphics_settings(w, h, bounces=4, spp=2)
    ref, st = oracle.render(spheres, materials, const, cam, gs4, threads=4)
    rng = np.random.default_rng(5)
    img, st2 = oracle.render(spheres, materials, set_cube(sd, first, random_rotation(rng)), cam, gs4, threads=4, textures=ts)
    assert st.rays == st2.rays and np.array_equal(bits(img), bits(ref))
    fn = lambda d: np.stack([0.5 + 0.4 * d[..., 0], 0.5 + 0.4 * d[..., 1] * d[..., 2], 1.0 + 0.5 * d[..., 2]], -1)
    ts2 = T.TextureSet(len(spheres))
    cube = ts2.add_cube(T.cube_from_function(64, fn))
    W, H = 256, 128
    v, u = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
    theta, phi = v * np.pi, (2 * u - 1) * np.pi
    lat = ts2.add_hdr_image(fn(np.stack([np.sin(theta) * np.sin(phi), np.cos(theta), np.sin(theta) * np.cos(phi)], -1)))
    m = random_rotation(rng)
    a, _ = oracle.render(spheres, materials, set_cube(sd, cube, m), cam, gs, threads=4, textures=ts2)
    b, _ = oracle.render(spheres, materials, set_env(sd, lat, m), cam, gs, threads=4, textures=ts2)
    assert np.allclose(a, b, rtol=0, atol=0.02)


# ---- GPU ---------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_ENV_SEEDS", "15"))))
def test_gpu_environment_map_matches_oracle(dxrs, host, oracle, renderer, seed):
    from dxrs_amd import textures as T
    rng = np.random.default_rng(9100 + seed)
    n = int(rng.choice([3, 8, 20, 500]))  # 500: BVH in global memory
    spheres, materials, ts = make_textured_scene(dxrs, rng, n, seed % 2)
    if seed % 3 == 0:  # the environment map alone: no sphere has maps
        ts = T.TextureSet(n)
    cube = seed % 5 in (1, 4)
    if cube:
        size = int(rng.integers(1, 9))
        idx = ts.add_cube([rng.uniform(0, 4, (size, size, 4)).astype(np.float32) for _ in range(6)])
    elif seed % 2:
        idx = ts.add_hdr_image(T.sky_latlong(128, 64, seed=seed))
    else:
        idx = ts.add_hdr_image(rng.uniform(0, 4, (int(rng.integers(1, 9)), int(rng.integers(1, 17)), 4)).astype(np.float32))
    matrix = [None, random_rotation(rng), rng.normal(size=(3, 3))][seed % 3]  # incl. a non-orthonormal transform (normalised after)
    sd = (set_cube if cube else set_env)(host.scene(dxrs.host.SCENE_SMALL)[2], idx, matrix)
    w, h = int(rng.choice([64, 97])), int(rng.choice([48, 61]))
    pos = (0.0, 0.5, -12.0) if seed % 3 else (0.2, 0.1, 0.0)
    cam = host.camera(w, h, position=pos, jitter_index=seed)
    gs = dxrs.types.graphics_settings(w, h, frame_index=seed, bounces=int(rng.choice([0, 2, 6])), spp=int(rng.choice([1, 3])), rr=bool(seed % 2), di=seed % 4 == 3)
    renderer.set_scene(spheres, materials, sd)
    renderer.set_textures(ts)
    renderer.set_camera(cam); renderer.set_constants(gs)
    img, st = renderer.render()
    ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8, textures=ts)
    assert st.rays == ost.rays
    assert count_mismatch(img, ref) == 0
    renderer.set_textures(None)


@pytest.mark.gpu
def test_gpu_environment_map_api(dxrs, host, oracle, renderer):
    from dxrs_amd import textures as T
    t = dxrs.types
    spheres, materials, sd = small_scene(dxrs, host)
    n = len(spheres)
    w, h = 128, 96
    cam, gs = host.camera(w, h), t.graphics_settings(w, h, bounces=3)
    renderer.set_camera(cam); renderer.set_constants(gs)
    # a cube map needs six square table entries of one size
    ts6 = T.TextureSet(n)
    ts6.add_cube([np.ones((2, 2, 3), np.float32)] * 5 + [np.ones((2, 3, 3), np.float32)])
    renderer.set_scene(spheres, materials, set_cube(sd, 0))
    renderer.set_textures(ts6)
    with pytest.raises(RuntimeError, match="square"):
        renderer.render()
    renderer.set_scene(spheres, materials, set_cube(sd, 1))
    renderer.set_textures(ts6)
    with pytest.raises(RuntimeError, match="six consecutive"):
        renderer.render()
    # a descriptor without a table fails at render time, loudly
    env_sd = set_env(sd, 0, rot_y(0.7))
    renderer.set_scene(spheres, materials, env_sd)
    with pytest.raises(RuntimeError, match="EnvironmentLightTextureDescriptor"):
        renderer.render()
    ts = T.TextureSet(n)
    env = T.sky_latlong(256, 128, seed=2)
    assert ts.add_hdr_image(env) == 0
    renderer.set_textures(ts)
    img, st = renderer.render()
    ref, ost = oracle.render(spheres, materials, env_sd, cam, gs, threads=8, textures=ts)
    assert st.rays == ost.rays and count_mismatch(img, ref) == 0
    # the table without per-object maps: object_textures = NULL through the raw C-ABI gives the same frame
    tex, n_tex, obj, rot = ts.as_ctypes()
    rc = renderer._lib.pt_set_textures(renderer._ctx, C.cast(tex, C.c_void_p), n_tex, None, None)
    assert rc == 0
    img2, _ = renderer.render()
    assert np.array_equal(bits(img2), bits(img))
    # a one-colour map == the constant-colour branch, bit for bit
    colour = np.array([0.25, 2.0, 1.5], np.float32)
    ts1 = T.TextureSet(n); ts1.add_hdr_image(np.tile(colour, (2, 2, 1)))
    renderer.set_textures(ts1)
    a, _ = renderer.render()
    const = copy.copy(sd)
    for k in range(3):
        const.EnvironmentLightColor[k] = float(colour[k])
    const.EnvironmentLightColor[3] = 1.0
    renderer.set_scene(spheres, materials, const)
    b, _ = renderer.render()
    assert np.array_equal(bits(a), bits(b))
    # a descriptor past the table
    renderer.set_scene(spheres, materials, set_env(sd, 3))
    renderer.set_textures(ts)
    with pytest.raises(RuntimeError, match="EnvironmentLightTextureDescriptor"):
        renderer.render()
    renderer.set_scene(spheres, materials, sd)
