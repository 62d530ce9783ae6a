"""Timeline of a short burst of C2 frames (the driver's 20-step run): when each frame's result becomes available relative to the first
submit, host submit times, and the drain after the last submit.  usage: python tools/experiments/burst.py [frames] [lanes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 1920, 1080
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=lanes)
r.set_scene(spheres, materials, sd)
gs = graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]
def frame(k):
    gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); r.render_device(bufs[k % lanes].data_ptr())
n_warm = 40
for k in range(n_warm):
    frame(k)
torch.cuda.synchronize()
calls = n_warm  # frame index = render-call count, so that buffer k % lanes stays on lane k % lanes in every repetition (a buffer that
                # moves to another lane between bursts makes render_common serialise the frames: the rotation contract of pt_render)
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record(ts)
    sub = []
    for k in range(n):
        frame(calls); calls += 1  # buffer = call count % lanes, the lane rotation of the renderer
        ev[k + 1].record(ts)
        sub.append(time.perf_counter() - t0)
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    done = [ev[0].elapsed_time(e) for e in ev[1:]]
    print(f"rep {rep}: {n} frames, {lanes} lanes: submit loop {t_sub * 1e3:.3f} ms, all done {t_all * 1e3:.3f} ms = {t_all / n * 1e3:.4f} ms/frame; "
          f"GPU: first frame done at {done[0]:.3f} ms, last at {done[-1]:.3f} ms; steady interval {(done[-1] - done[n // 2]) / (n - 1 - n // 2):.4f} ms")
    print("   submit times (ms):", " ".join(f"{s * 1e3:.2f}" for s in sub[:8]), "...", f"{sub[-1] * 1e3:.2f}")
    print("   done times   (ms):", " ".join(f"{d:.2f}" for d in done[:8]), "...", f"{done[-1]:.2f}")
r.close()
