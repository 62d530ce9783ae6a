"""What the textured kernels cost on C2: untextured kernels / textured kernels with no sphere carrying a map (the price of the code:
registers, spills) / the demo's three textured objects.  usage: python tools/experiments/texcost.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings
from dxrs_amd.textures import NO_TEXTURE

w, h, lanes = 1920, 1080, 3
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]

def run(tag, tex):
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=lanes)
        r.set_scene(spheres, materials, sd)
        if tex is not None:
            r.set_textures(tex)
        gs = graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
        bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
        def frame(k):
            gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); r.render_device(bufs[k % lanes].data_ptr())
        for k in range(30): frame(k)
        res = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(300): frame(30 + rep * 300 + k)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 300 * 1e3)
        print(f"{tag:40s} " + " ".join(f"{x:.4f}" for x in res) + " ms/frame")
        r.close()

run("untextured kernels", None)
tex = host.demo_textures(0, 0.0, textured=True)
empty = host.demo_textures(0, 0.0, textured=True)
empty.maps[:] = NO_TEXTURE
run("textured kernels, no sphere has a map", empty)
run("demo textures (3 objects)", tex)
