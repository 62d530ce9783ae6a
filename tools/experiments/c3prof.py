import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H, SPP = 3840, 2160, 16
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
r = dxrs_amd.Renderer(stream=ts.cuda_stream, frames_in_flight=1)
r.set_scene(s, m, sd)
gs = dxrs_amd.types.graphics_settings(W, H, spp=SPP); r.set_constants(gs)
cam = host.camera(W, H)
buf = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
def step(k):
    gs.FrameIndex = k; r.set_camera(cam); r.set_constants(gs); r.render_device(buf.data_ptr())
for k in range(3): step(k)
torch.cuda.synchronize()
r.totals(reset=True)
r.set_profiling(True)
N = 5
t0 = time.perf_counter()
for k in range(N): step(3 + k)
torch.cuda.synchronize()
t1 = time.perf_counter()
p = r.profile(reset=True); tot = r.totals()
print("frame %.2f ms; rays/frame %.1f M; compacting launches/frame %.1f avg %.1f us (sum %.2f ms); loop %.1f launches avg %.1f us" % (
    (t1 - t0) / N * 1e3, tot.rays / N / 1e6, p.traverse_launches / N, p.ms_traverse / max(p.traverse_launches, 1) * 1e3, p.ms_traverse / N,
    p.tail_launches / N, p.ms_tail / max(p.tail_launches, 1) * 1e3))
os.environ["PT_DEBUG_COUNTS"] = "1"
img, st = r.render()
