import numpy as np


def rel_l2(a, b):
    """relative L2 error of the rgb channels of a against reference b"""
    a = np.asarray(a, dtype=np.float64)[..., :3]
    b = np.asarray(b, dtype=np.float64)[..., :3]
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-300))


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def count_mismatch(a, b):
    """number of pixels whose rgb differs in any bit"""
    return int((bits(a)[..., :3] != bits(b)[..., :3]).any(axis=-1).sum())


def render_rested(renderer, rect=None, expect_beams=None):
    """Render the current view three times: the first frame traverses the BVH for every primary ray, the second (the view has
    rested) starts the primary-beam build, the third takes its primary candidates from the beam lists (DESIGN.md "Primary
    beams").  All three must agree bit for bit; returns the third (image, stats)."""
    first, st1 = renderer.render(rect)  # (a view that already rested under other settings may use its lists here: they do not depend on spp / bounces)
    renderer.render(rect)
    img, st = renderer.render(rect)
    if expect_beams is not None:
        assert bool(st.beams_used) == expect_beams
    assert st.rays == st1.rays
    assert np.array_equal(bits(img), bits(first)), "beam-list frame differs from the per-ray traversal frame"
    return img, st
