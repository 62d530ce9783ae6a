"""Latency anatomy of the looping pass at C2: its duration (one frame in flight, exclusive) as a function of the bounce limit --
fixed cost (launch, BVH staging, segment prefix sum) vs cost per dependent bounce.  usage: python tools/experiments/looplat.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings

w, h = 1920, 1080
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
r = dxrs_amd.Renderer(device=0, frames_in_flight=1)
r.set_scene(spheres, materials, sd)
buf = torch.empty((h * w, 4), dtype=torch.float32, device="cuda")
for bounces in (1, 2, 3, 4, 5, 6, 8, 12):
    gs = graphics_settings(w, h, frame_index=0, bounces=bounces, spp=1)
    for k in range(6):
        gs.FrameIndex = k; r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs); r.render_device(buf.data_ptr())
    r.set_profiling(True)
    n = 30
    for k in range(n):
        gs.FrameIndex = k; r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs); r.render_device(buf.data_ptr())
    p = r.profile(reset=True)
    r.set_profiling(False)
    q = r.queue_sizes()
    print(f"bounces {bounces:2d}: primary {p.ms_traverse / n * 1e3:6.1f} us  loop {p.ms_tail / max(p.tail_launches, 1) * 1e3:6.1f} us ({p.tail_launches / n:.0f}/frame)  queue sizes {q[:3]}")
r.close()
