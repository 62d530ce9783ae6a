#!/usr/bin/env python3
"""Render the demo scene on the GPU and write a PNG through the product's own display path: N frames of the path tracer
with jittered cameras -> pt_accumulate (running mean) -> pt_tonemap (ACES filmic + sRGB, the reference's SDR default) ->
R8G8B8A8.  Viewer convenience; the measured output of the hot path is the fp32 HDR radiance buffer.

    python tools/render_png.py out.png [--width 1280 --height 720 --spp 8 --frames 16 --bounces 8 --time 0.0 --textures
                                        --texture-dir /path/to/Assets/Textures]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device buffers only)
import dxrs_amd_loader  # noqa: E402,F401
import dxrs_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16, help="jittered frames accumulated by pt_accumulate")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--time", type=float, default=0.0, help="simulation time of the closed-form motion")
    ap.add_argument("--textures", action="store_true", help="textured Alien-Metal / Moon / Earth (procedural stand-ins)")
    ap.add_argument("--env-map", action="store_true", help="lat-long HDR environment light (MyScene.ixx:94-95; procedural stand-in) instead of the sky")
    ap.add_argument("--texture-dir", default=None, help="directory of decoded images (<stem>.ptex, e.g. tests/golden/textures = the reference's Assets/Textures): use them instead of the stand-ins")
    ap.add_argument("--operator", choices=["saturate", "reinhard", "aces"], default="aces")
    ap.add_argument("--exposure", type=float, default=0.0, help="stops")
    args = ap.parse_args()
    from PIL import Image

    t = dxrs_amd.types
    host = dxrs_amd.load_host()
    spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
    if args.time:
        spheres = host.scene_at_time(0, args.time)
    torch.cuda.init()
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    r = dxrs_amd.Renderer(stream=stream.cuda_stream)
    textured = bool(args.textures or args.texture_dir)
    if textured or args.env_map:
        # --texture-dir: decoded images (<stem>.ptex; tests/golden/textures holds the reference's own assets) through the host mirror's loader
        ts, sd_env = host.demo_textures(0, args.time, textured=textured, environment_map=args.env_map, return_scene_data=True, texture_dir=args.texture_dir)
        if args.env_map:
            sd = sd_env
    r.set_scene(spheres, materials, sd)
    if textured or args.env_map:
        r.set_textures(ts)
    w, h, n = args.width, args.height, args.width * args.height
    gs = t.graphics_settings(w, h, bounces=args.bounces, spp=args.spp)
    frame = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    accum = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    ldr = torch.empty(n, dtype=torch.int32, device="cuda")
    for k in range(args.frames):
        gs.FrameIndex = k
        r.set_camera(host.camera(w, h, jitter_index=k, jitter_count=max(args.frames, 8)))
        r.set_constants(gs)
        r.render_device(frame.data_ptr())
        r.accumulate(accum.data_ptr(), frame.data_ptr(), n, k)
    op = {"saturate": t.TONE_SATURATE, "reinhard": t.TONE_REINHARD, "aces": t.TONE_ACES_FILMIC}[args.operator]
    r.tonemap(accum.data_ptr(), n, t.tonemap_params(op, t.TRANSFER_SRGB, args.exposure), ldr.data_ptr())
    r.synchronize()
    rgba = ldr.cpu().numpy().view(np.uint8).reshape(h, w, 4)
    Image.fromarray(rgba[..., :3]).save(args.out)
    tot = r.totals()
    print(f"{args.frames} frames x {args.spp} spp, {tot.rays} rays -> {args.out}")
    r.close()


if __name__ == "__main__":
    main()
