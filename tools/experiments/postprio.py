"""Does rendering on highest-priority lanes starve the caller's post-processing?  C2 frames, each followed on the caller's stream by
pt_accumulate + pt_tonemap of that frame (what a viewer does).  usage: PT_LANE_PRIORITY=0|1 python tools/experiments/postprio.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings, tonemap_params
w, h, lanes = 1920, 1080, 3
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=lanes)
r.set_scene(spheres, materials, sd)
gs = graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
accum = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
ldr = torch.zeros((h * w,), dtype=torch.int32, device="cuda")
tp = tonemap_params()
def frame(k, post):
    gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); r.render_device(bufs[k % lanes].data_ptr())
    if post:
        r.accumulate(accum.data_ptr(), bufs[k % lanes].data_ptr(), h * w, k % 64)
        r.tonemap(accum.data_ptr(), h * w, tp, ldr.data_ptr())
for post in (False, True):
    for k in range(30): frame(k, post)
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(300): frame(30 + rep * 300 + k, post)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 300 * 1e3)
    print(f"PT_LANE_PRIORITY={os.environ.get('PT_LANE_PRIORITY', 'default(1)')} post={post}: " + " ".join(f"{x:.4f}" for x in res) + " ms/frame")
r.close()
