"""Generates the committed golden fixtures from the CPU oracle (there is no runnable reference: SURVEY F1/F2/F5, so
these are self-generated; "parity unpinned").  Run from the repo root:  python tests/golden/make_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import dxrs_amd_loader  # noqa: E402,F401
import dxrs_amd  # noqa: E402
from oracle.binding import load_oracle  # noqa: E402


def main():
    oracle = load_oracle()
    host = dxrs_amd.load_host()
    lib = oracle.lib
    seeds = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1919, 1079, 0), (3839, 2159, 7), (65535, 65535, 0xFFFFFFFF), (123, 456, 789)]
    rng = np.zeros((8, 17), dtype=np.uint32)
    for i, (x, y, f) in enumerate(seeds):
        s = C.c_uint32(lib.oracle_rng_init(x, y, f)); rng[i, 0] = s.value
        for k in range(16):
            rng[i, 1 + k] = lib.oracle_rng_next(C.byref(s))
    np.save(os.path.join(HERE, "rng_streams.npy"), rng)

    halton = np.array([[lib.oracle_halton(i, b) for b in (2, 3, 5)] for i in range(1, 65)], dtype=np.float32)
    np.save(os.path.join(HERE, "halton_1_64.npy"), halton)

    sys.path.insert(0, os.path.dirname(HERE))
    import golden_cases
    for c in golden_cases.cases(dxrs_amd, host):
        img, _ = oracle.render(c["spheres"], c["materials"], c["sd"], c["cam"], c["gs"], rect=c["rect"], threads=8, textures=c["textures"])
        np.save(os.path.join(HERE, c["file"]), img)
    src, dst, params = golden_cases.tonemap_case(dxrs_amd)
    np.save(os.path.join(HERE, dst), oracle.tonemap(np.load(os.path.join(HERE, src)), params))
    spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_SMALL, seed=0)
    spheres2, materials2, sd2 = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
    # the scenes themselves (so a change of the scene generator is caught separately from a change of the estimator)
    np.save(os.path.join(HERE, "scene_small_seed0_spheres.npy"), spheres)
    np.save(os.path.join(HERE, "scene_demo_seed0_spheres.npy"), spheres2)
    np.save(os.path.join(HERE, "scene_demo_seed0_materials.npy"), materials2)
    print("golden fixtures written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npy")))


if __name__ == "__main__":
    main()
