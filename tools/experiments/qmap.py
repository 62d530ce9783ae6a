"""How the C2 frame time depends on the hardware queues the lane streams land on: K dummy streams are created (and used once) before
the renderer, which shifts the runtime's round-robin of streams over its hardware queues.  usage: python tools/experiments/qmap.py [lanes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings
w, h = 1920, 1080
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]
tex = host.demo_textures(0, 0.0, textured=True)
dummies = []
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
def run(textured):
    r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=lanes)
    r.set_scene(spheres, materials, sd)
    if textured: r.set_textures(tex)
    gs = graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
    bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
    def frame(k):
        gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); r.render_device(bufs[k % lanes].data_ptr())
    for k in range(30): frame(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(300): frame(30 + k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e3
    r.close()
    return dt
for k in range(9):
    a = run(False); b = run(True)
    print(f"{len(dummies)} dummy streams before the context: untextured {a:.4f}  textured {b:.4f} ms/frame", flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    dummies.append(s)
