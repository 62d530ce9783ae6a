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
