"""Randomised scenes through the whole pipeline vs the CPU oracle: overlapping / nested / coincident spheres (tie-break by
id), the camera inside spheres, extreme radii, and material corner cases (roughness 0 and 1, IOR 1, metallic +
transmission, strong emitters, black and over-range base colours).  Bit-exact, like every other parity test."""
import os

import numpy as np
import pytest

from util import count_mismatch, render_rested

pytestmark = pytest.mark.gpu


def random_scene(dxrs, rng, n):
    s = np.zeros(n, dtype=dxrs.SPHERE_DTYPE)
    style = rng.integers(0, 4)
    if style == 0:      # loose cloud in front of the camera
        s["cx"], s["cy"], s["cz"] = rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-8, 8, n)
        s["r"] = np.exp(rng.uniform(np.log(0.05), np.log(3.0), n))
    elif style == 1:    # heavy overlap + nesting around the origin, camera possibly inside
        s["cx"], s["cy"], s["cz"] = rng.normal(0, 1.5, n), rng.normal(0, 1.5, n), rng.normal(-12, 3.0, n)
        s["r"] = np.exp(rng.uniform(np.log(0.2), np.log(6.0), n))
    elif style == 2:    # tiny spheres next to a huge one (the demo's ground situation)
        s["cx"], s["cy"], s["cz"] = rng.uniform(-5, 5, n), rng.uniform(0, 1, n), rng.uniform(-5, 5, n)
        s["r"] = rng.uniform(0.02, 0.3, n)
        s[0] = (0, -1000.2, 0, 1000.0)
    else:               # coincident duplicates: the closest-hit tie must go to the lowest id
        base = max(1, n // 3)
        s["cx"][:base], s["cy"][:base], s["cz"][:base] = rng.uniform(-4, 4, base), rng.uniform(-3, 3, base), rng.uniform(-6, 6, base)
        s["r"][:base] = rng.uniform(0.3, 2.0, base)
        for i in range(base, n):
            s[i] = s[rng.integers(0, base)]
    m = dxrs.types.default_material(n)
    m["BaseColor"][:, :3] = rng.choice([0.0, 1.0, 0.5, 1.5], (n, 3)) * rng.random((n, 3)) ** 0.5
    m["Metallic"] = rng.choice([0.0, 1.0, 0.3], n)
    m["Roughness"] = rng.choice([0.0, 1.0, 0.05, 0.5], n)
    m["Transmission"] = rng.choice([0.0, 1.0, 0.5], n)
    m["IOR"] = rng.choice([1.5, 1.0, 1.33, 2.4, 1.0001], n)
    emit = rng.random(n) < 0.25
    m["EmissiveStrength"][emit] = rng.uniform(0.5, 50.0, emit.sum())
    m["EmissiveColor"][emit] = rng.random((emit.sum(), 3))
    if rng.random() < 0.5:  # alpha-tested hits (spec S10) in half of the scenes: Mask / Blend spheres above, at and below their cut-off
        a = rng.random(n) < 0.3
        m["AlphaMode"][a] = rng.choice([1, 2], a.sum())
        m["BaseColor"][a, 3] = rng.choice([0.0, 0.3, 0.5, 0.9, 1.0], a.sum())
        m["AlphaCutoff"][a] = rng.choice([0.5, 0.25, 1.0], a.sum())
    return s, m


# PT_FUZZ_SEEDS=N widens the sweep for a soak run (the default 96 keeps the suite short)
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_SEEDS", "96"))))
def test_random_scene_matches_oracle(dxrs, host, oracle, renderer, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 200]))
    spheres, materials = random_scene(dxrs, rng, n)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    if seed % 3 == 0:  # constant environment instead of the sky
        sd.EnvironmentLightColor[0], sd.EnvironmentLightColor[1], sd.EnvironmentLightColor[2], sd.EnvironmentLightColor[3] = 0.7, 0.8, 1.1, 1.0
    w, h = int(rng.choice([48, 64, 81])), int(rng.choice([40, 48, 57]))
    bounces, spp = int(rng.choice([0, 1, 3, 6, 12])), int(rng.choice([1, 2, 5]))
    pos = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -12.0) if seed % 4 else (0.1, 0.2, 0.3)  # sometimes inside the cluster
    cam = host.camera(w, h, position=pos, look_at=(0.0, 0.0, 0.0) if seed % 2 else None, jitter_index=seed)
    gs = dxrs.types.graphics_settings(w, h, frame_index=seed * 7919, bounces=bounces, spp=spp, rr=bool(seed % 5))
    renderer.set_scene(spheres, materials, sd)
    renderer.set_camera(cam)
    renderer.set_constants(gs)
    img, st = render_rested(renderer, expect_beams=n > 1)  # first frame: per-ray traversal; third: primary-beam lists
    ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
    assert st.rays == ost.rays
    assert count_mismatch(img, ref) == 0
    assert np.isfinite(img).all()


# scenes too large for the LDS-resident path: global-memory BVH, split traverse/shade schedule
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_LARGE_SEEDS", "12"))))
def test_random_large_scene_matches_oracle(dxrs, host, oracle, renderer, seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([450, 1000, 4000]))
    spheres, materials = random_scene(dxrs, rng, n)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    w, h = int(rng.choice([96, 129])), int(rng.choice([64, 75]))
    bounces, spp = int(rng.choice([1, 4, 8])), int(rng.choice([1, 3]))
    pos = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -12.0) if seed % 4 else (0.1, 0.2, 0.3)
    cam = host.camera(w, h, position=pos, look_at=(0.0, 0.0, 0.0) if seed % 2 else None, jitter_index=seed)
    gs = dxrs.types.graphics_settings(w, h, frame_index=seed * 104729, bounces=bounces, spp=spp, rr=bool(seed % 3))
    renderer.set_scene(spheres, materials, sd)
    renderer.set_camera(cam)
    renderer.set_constants(gs)
    img, st = renderer.render()
    ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
    assert st.rays == ost.rays
    assert count_mismatch(img, ref) == 0


# sphere counts around the limits of the LDS-resident path: the scene copy (84 B per sphere) plus the per-lane stacks cross the
# 64 KB default LDS limit of a workgroup (static + dynamic) near 600-700 spheres and the residency budget shortly after
@pytest.mark.parametrize("n", [520, 560, 600, 620, 640, 660, 680, 700, 740, 780, 820])
def test_scene_sizes_around_the_lds_limits(dxrs, host, oracle, renderer, n):
    rng = np.random.default_rng(9000 + n)
    spheres, materials = random_scene(dxrs, rng, n)
    sd = host.scene(dxrs.host.SCENE_SMALL)[2]
    w, h = 96, 64
    cam = host.camera(w, h, position=(0.3, 0.2, -12.0), jitter_index=n)
    renderer.set_scene(spheres, materials, sd)
    renderer.set_camera(cam)
    for spp, bounces in ((1, 5), (3, 2)):
        gs = dxrs.types.graphics_settings(w, h, frame_index=n, bounces=bounces, spp=spp)
        renderer.set_constants(gs)
        img, st = render_rested(renderer)
        ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
        assert st.rays == ost.rays
        assert count_mismatch(img, ref) == 0
