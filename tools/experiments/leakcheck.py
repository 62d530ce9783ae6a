"""Context life cycle: create / use every feature / destroy, many times; free device memory must come back.
usage: python tools/experiments/leakcheck.py [iterations]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
big = host.scene(dxrs_amd.host.SCENE_PROCEDURAL, seed=1, count=20000)
tex, sd_env = host.demo_textures(0, 0.0, textured=True, environment_map=True, return_scene_data=True)
w, h = 320, 200
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
torch.cuda.synchronize()
free0 = None
for it in range(n_iter):
    r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=3)
    r.set_scene(spheres, materials, sd_env); r.set_textures(tex)
    gs = graphics_settings(w, h, frame_index=it, bounces=4, spp=2, di=True)
    r.set_camera(host.camera(w, h)); r.set_constants(gs)
    for k in range(4):
        r.render_device(bufs[k % 3].data_ptr())
    r.update_spheres(host.scene_at_time(0, 0.1 * it))
    r.render_device(bufs[1].data_ptr())
    r.set_scene(*big)                       # global-memory BVH, wide nodes, split schedule
    r.set_constants(graphics_settings(w, h, frame_index=it, bounces=3, spp=1))
    r.render_device(bufs[2].data_ptr())
    r.set_partition(1, 3)
    r.render_tiles(bufs[0].data_ptr())
    r.set_scene(spheres[:0], materials[:0], sd)
    r.set_partition(0, 1)
    r.render_device(bufs[0].data_ptr())
    r.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 2:
        free0 = free   # after the first iterations: code objects, allocator pools and caches of the runtime are in place
    if it >= 2 and (it % 8 == 2 or it == n_iter - 1):
        print(f"iteration {it:3d}: free device memory {free / 2**20:10.1f} MiB (drift since iteration 2: {(free0 - free) / 2**20:+.1f} MiB)", flush=True)
