#!/usr/bin/env python3
"""Render the demo scene on the GPU and write a tone-mapped PNG (viewer convenience; the tone map -- Reinhard + gamma 2.2 in
numpy -- is NOT part of the measured path, the product's output is the fp32 HDR radiance buffer).

    python tools/render_png.py out.png [--width 1280 --height 720 --spp 64 --bounces 8 --time 0.0]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dxrs_amd_loader  # noqa: E402,F401
import dxrs_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--time", type=float, default=0.0, help="simulation time of the closed-form motion")
    args = ap.parse_args()
    from PIL import Image

    host = dxrs_amd.load_host()
    spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
    if args.time:
        spheres = host.scene_at_time(0, args.time)
    r = dxrs_amd.Renderer()
    r.set_scene(spheres, materials, sd)
    r.set_camera(host.camera(args.width, args.height, jitter=False))
    r.set_constants(dxrs_amd.types.graphics_settings(args.width, args.height, bounces=args.bounces, spp=args.spp))
    img, st = r.render()
    print(f"{st.rays} rays in {st.ms_total:.2f} ms ({st.rays / st.ms_total / 1e3:.0f} Mrays/s)")
    x = np.clip(img[..., :3], 0, None)
    x = np.clip(x / (1 + x), 0, 1) ** (1 / 2.2)
    Image.fromarray((x * 255 + 0.5).astype(np.uint8)).save(args.out)
    r.close()


if __name__ == "__main__":
    main()
