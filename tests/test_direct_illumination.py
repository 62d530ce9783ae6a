"""Row N4 -- sphere-light direct illumination of the primary surface (IsDIEnabled = 1), the build's stand-in for the
reference's RTXDI passes (Raytracing.hlsl:150-163, 302, 381; LightPreparation.ixx:52-70).  CPU: the cone sampler against
analytic answers and the device header bit for bit; the estimator is unbiased -- averaged over many frames, DI on and DI off
converge to the same image (DI only moves the emitters' first-bounce contribution from a random hit to an explicit sample).
GPU: frames with DI through the C-ABI against the oracle, bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.binding import declare_leaf_api
from util import count_mismatch, render_rested

HERE = os.path.dirname(os.path.abspath(__file__))
PF = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def dev():
    lib = C.CDLL(os.path.join(HERE, "hostshim", "libdevmath_host.so"))
    declare_leaf_api(lib, "dev_")
    return lib


def fa(*v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def cone(fn, P, Cc, r, u1, u2):
    L, ip = (C.c_float * 3)(), C.c_float()
    ok = fn(fa(*P), fa(*Cc), C.c_float(r), C.c_float(u1), C.c_float(u2), L, C.byref(ip))
    return ok, np.array(L[:], dtype=np.float32), np.float32(ip.value)


def test_cone_sampler(oracle, dev):
    rng = np.random.default_rng(0)
    f = oracle.lib.oracle_sample_sphere_cone
    # inside / on the emitter: no sample
    assert cone(f, (0, 0, 0), (0.1, 0, 0), 1.0, 0.5, 0.5)[0] == 0
    for _ in range(2000):
        P, Cc = rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3)
        r = float(np.exp(rng.uniform(np.log(1e-3), np.log(3.0))))
        u1, u2 = 1 - rng.random(), 1 - rng.random()  # (0, 1]
        ok, L, ip = cone(f, P, Cc, r, u1, u2)
        ok2, L2, ip2 = cone(dev.dev_sample_sphere_cone, P, Cc, r, u1, u2)
        assert ok == ok2 and np.array_equal(L.view(np.uint32), L2.view(np.uint32)) and ip.view(np.uint32) == ip2.view(np.uint32)
        d = np.linalg.norm(Cc - P)
        if not ok:
            assert d <= r * (1 + 1e-5)
            continue
        assert abs(np.linalg.norm(L) - 1) < 2e-6
        # the direction lies inside the cone: the ray P + tL passes within r of the centre
        w = (Cc - P).astype(np.float64)
        miss = np.linalg.norm(w - np.dot(w, L.astype(np.float64)) * L.astype(np.float64))
        assert miss <= r * (1 + 2e-4) + 1e-6
        # 1 / pdf = solid angle of the cone = 2 pi (1 - cos theta_max), also for tiny far emitters
        omega = 2 * np.pi * (1 - np.sqrt(max(0.0, 1 - (r / d) ** 2))) if r / d > 1e-3 else np.pi * (r / d) ** 2
        assert abs(ip - omega) <= 2e-5 * omega + 1e-12
    # u1 -> 0 is the axis, u1 = 1 the rim
    _, L, _ = cone(f, (0, 0, 0), (0, 0, 4), 1.0, 1e-7, 0.3)
    assert L[2] > 0.999999
    _, L, _ = cone(f, (0, 0, 0), (0, 0, 4), 1.0, 1.0, 0.3)
    assert abs(L[2] - np.sqrt(1 - 1 / 16)) < 1e-6


def lit_scene(dxrs, n_lights=3, glass=False):
    """a diffuse floor (big sphere), a few diffuse / metal spheres and bright emitters above them"""
    t = dxrs.types
    n = 6 + n_lights
    s = np.zeros(n, dtype=dxrs.SPHERE_DTYPE)
    m = t.default_material(n)
    s[0] = (0, -100.5, 0, 100.0); m["BaseColor"][0, :3] = (0.7, 0.7, 0.7); m["Roughness"][0] = 1.0
    pos = [(-2.2, 0.5, 0), (0, 0.5, 0.5), (2.2, 0.5, 0), (-1.0, 0.3, -2.0), (1.2, 0.3, -2.2)]
    for i, p in enumerate(pos):
        s[1 + i] = (*p, 0.8 if i < 3 else 0.5)
        m["BaseColor"][1 + i, :3] = [(0.8, 0.3, 0.3), (0.3, 0.8, 0.3), (0.3, 0.3, 0.8), (0.9, 0.9, 0.9), (0.8, 0.7, 0.2)][i]
        m["Roughness"][1 + i] = [1.0, 0.6, 0.3, 1.0, 0.4][i]
        m["Metallic"][1 + i] = [0, 0, 1, 0, 1][i]
    if glass:
        m["Transmission"][2] = 1.0; m["Roughness"][2] = 0.0
    for k in range(n_lights):
        s[6 + k] = (-3 + 3 * k, 3.0 + 0.5 * k, -1.0 + k, 0.35)
        m["BaseColor"][6 + k, :3] = 0.0
        m["EmissiveStrength"][6 + k] = 25.0
        m["EmissiveColor"][6 + k] = [(1, 0.8, 0.6), (0.6, 0.8, 1), (0.8, 1, 0.7)][k % 3]
    return s, m


def test_oracle_di_is_unbiased(dxrs, host, oracle):
    """mean over frames: DI on == DI off (within Monte-Carlo error) on a scene lit by sphere emitters only, and much less noisy"""
    t = dxrs.types
    spheres, materials = lit_scene(dxrs)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    sd.EnvironmentLightColor[0] = sd.EnvironmentLightColor[1] = sd.EnvironmentLightColor[2] = 0.0; sd.EnvironmentLightColor[3] = 1.0  # black environment
    w, h, frames = 48, 32, 96
    acc = {False: np.zeros((h, w, 3)), True: np.zeros((h, w, 3))}
    var = {False: 0.0, True: 0.0}
    for di in (False, True):
        imgs = []
        for k in range(frames):
            gs = t.graphics_settings(w, h, frame_index=k, bounces=1, spp=4, rr=False, di=di)  # bounces = 1: direct light only
            img, _ = oracle.render(spheres, materials, sd, host.camera(w, h, position=(0, 1.5, -7), jitter=False), gs, threads=8)
            imgs.append(img[..., :3].astype(np.float64))
        imgs = np.stack(imgs)
        acc[di], var[di] = imgs.mean(0), imgs.var(0).mean()
    lit = acc[False].sum(-1) > 0.02
    rel = np.abs(acc[True] - acc[False]).sum(-1)[lit].mean() / acc[False].sum(-1)[lit].mean()
    assert lit.mean() > 0.3 and rel < 0.08, rel
    assert abs(acc[True].mean() / acc[False].mean() - 1) < 0.03      # same energy
    assert var[True] < 0.25 * var[False]                              # and far less variance at equal cost


def test_oracle_di_off_is_the_default_path(dxrs, host, oracle):
    """IsDIEnabled on a scene without emitters changes nothing; with emitters the primary-only (Bounces = 0) frame gains DI"""
    t = dxrs.types
    spheres, materials, sd = host.scene(dxrs.host.SCENE_SMALL, seed=0)
    no_light = materials.copy(); no_light["EmissiveStrength"] = 0
    w, h = 48, 32
    cam = host.camera(w, h)
    a, sa = oracle.render(spheres, no_light, sd, cam, t.graphics_settings(w, h, bounces=3), threads=4)
    b, sb = oracle.render(spheres, no_light, sd, cam, t.graphics_settings(w, h, bounces=3, di=True), threads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.rays == sb.rays
    s2, m2 = lit_scene(dxrs)
    cam2 = host.camera(w, h, position=(0, 1.5, -7))
    c, _ = oracle.render(s2, m2, sd, cam2, t.graphics_settings(w, h, bounces=0), threads=4)
    d, sdd = oracle.render(s2, m2, sd, cam2, t.graphics_settings(w, h, bounces=0, di=True), threads=4)
    assert (d[..., :3] >= c[..., :3]).all() and d[..., :3].sum() > c[..., :3].sum() + 1.0
    assert w * h < sdd.rays <= 2 * w * h  # the primary ray (shared by the DI pass and the bounce loop, as the G-buffer is in the reference) + at most one shadow ray per pixel


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_DI_SEEDS", "10"))))
def test_gpu_di_matches_oracle(dxrs, host, oracle, renderer, seed):
    from test_textures import make_textured_scene
    t = dxrs.types
    rng = np.random.default_rng(9000 + seed)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    ts = None
    if seed % 3 == 0:
        spheres, materials = lit_scene(dxrs, n_lights=int(rng.integers(1, 4)), glass=bool(seed % 2))
        pos = (0, 1.5, -7)
    elif seed % 3 == 1:  # random scenes with random emitters + textures (emissive / normal maps on the primary surface)
        n = int(rng.choice([8, 20, 500]))
        spheres, materials, ts = make_textured_scene(dxrs, rng, n, 1)
        pos = (0.0, 0.5, -12.0)
    else:  # the demo-like small scene: emitters among many spheres, camera jittered
        spheres, materials, _ = host.scene(dxrs.host.SCENE_DEMO, seed=seed)
        pick = rng.choice(len(spheres) - 3, 12, replace=False)
        materials["EmissiveStrength"][pick] = rng.uniform(2, 30, 12); materials["EmissiveColor"][pick] = rng.random((12, 3))
        pos = (0, 0, -15)
    w, h = int(rng.choice([64, 97])), int(rng.choice([48, 61]))
    cam = host.camera(w, h, position=pos, jitter_index=seed)
    gs = t.graphics_settings(w, h, frame_index=seed * 31, bounces=int(rng.choice([0, 1, 3, 6])), spp=int(rng.choice([1, 3])), rr=bool(seed % 2), di=True)
    renderer.set_scene(spheres, materials, sd)
    renderer.set_textures(ts)
    renderer.set_camera(cam); renderer.set_constants(gs)
    img, st = render_rested(renderer)  # (the third frame of the view takes its primary candidates from the beam lists)
    ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8, textures=ts)
    assert st.rays == ost.rays
    assert count_mismatch(img, ref) == 0
    renderer.set_textures(None)


def _textured_emitter_scene(dxrs):
    """lit_scene with ONE emitter whose emissive map is a two-tone pattern (half of it black): the light the scene receives from
    it depends on which part of the emitter a shadow ray reaches"""
    from dxrs_amd import textures as T
    spheres, materials = lit_scene(dxrs, n_lights=1)
    ts = T.TextureSet(len(spheres))
    img = np.zeros((8, 16, 4), dtype=np.uint8)
    img[..., 3] = 255
    img[:, :8, :3] = 255            # u < 0.5: full emission; u >= 0.5: none
    tex = ts.add_image(img, srgb=False)
    ts.assign(6, dxrs.types.TEXTURE_MAP_EMISSIVE_COLOR, tex)
    return spheres, materials, ts


def test_oracle_di_with_a_textured_emitter_is_unbiased(dxrs, host, oracle):
    """the estimate must use the emitter's radiance AFTER EvaluateMaterial (emissive map), the same radiance whose
    first-bounce contribution shade drops -- otherwise DI on / off disagree in the mean.  Direct light only (Bounces = 1)."""
    t = dxrs.types
    spheres, materials, ts = _textured_emitter_scene(dxrs)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    sd.EnvironmentLightColor[0] = sd.EnvironmentLightColor[1] = sd.EnvironmentLightColor[2] = 0.0; sd.EnvironmentLightColor[3] = 1.0
    w, h, frames = 48, 32, 64
    mean = {}
    for di in (False, True):
        acc = np.zeros((h, w, 3))
        for k in range(frames):
            gs = t.graphics_settings(w, h, frame_index=k, bounces=1, spp=4, rr=False, di=di)
            img, _ = oracle.render(spheres, materials, sd, host.camera(w, h, position=(0, 1.5, -7), jitter=False), gs, threads=8, textures=ts)
            acc += img[..., :3]
        mean[di] = acc / frames
    # the map halves the emitter: against the untextured emitter the scene receives clearly less light
    full, _ = oracle.render(spheres, materials, sd, host.camera(w, h, position=(0, 1.5, -7), jitter=False),
                            t.graphics_settings(w, h, frame_index=0, bounces=1, spp=64, rr=False, di=True), threads=8)
    floor = (slice(20, 32), slice(8, 40))  # the lit floor in front of the spheres, no emitter pixels
    assert mean[True][floor].mean() < 0.8 * full[..., :3][floor].mean()
    assert abs(mean[True][floor].mean() / mean[False][floor].mean() - 1) < 0.06  # same energy with and without the estimator


@pytest.mark.gpu
def test_gpu_di_with_a_textured_emitter_matches_oracle(dxrs, host, oracle, renderer):
    t = dxrs.types
    spheres, materials, ts = _textured_emitter_scene(dxrs)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    w, h = 96, 64
    cam = host.camera(w, h, position=(0, 1.5, -7), jitter_index=4)
    renderer.set_scene(spheres, materials, sd)
    renderer.set_textures(ts)
    renderer.set_camera(cam)
    try:
        for spp, bounces in ((1, 3), (3, 0), (2, 5)):
            gs = t.graphics_settings(w, h, frame_index=17, bounces=bounces, spp=spp, di=True)
            renderer.set_constants(gs)
            img, st = render_rested(renderer)
            ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8, textures=ts)
            assert st.rays == ost.rays and count_mismatch(img, ref) == 0
    finally:
        renderer.set_textures(None)
