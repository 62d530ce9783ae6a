import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
for (W, H) in ((640, 384), (1920, 1080)):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    r = dxrs_amd.Renderer(stream=ts.cuda_stream, frames_in_flight=1)
    r.set_scene(s, m, sd)
    gs = dxrs_amd.types.graphics_settings(W, H); r.set_constants(gs)
    cams = [host.camera(W, H, jitter_index=k) for k in range(8)]
    buf = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
    def step(k):
        gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); r.render_device(buf.data_ptr())
    for k in range(30): step(k)
    torch.cuda.synchronize()
    r.set_profiling(True)
    N = 100
    t0 = time.perf_counter()
    for k in range(N): step(30 + k)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    p = r.profile(reset=True)
    print(W, H, "frame %.1f us (with events)  compacting launches/frame %.1f avg %.1f us   loop avg %.1f us   queue sizes %s" % (
        (t1 - t0) / N * 1e6, p.traverse_launches / N, p.ms_traverse / max(p.traverse_launches, 1) * 1e3, p.ms_tail / max(p.tail_launches, 1) * 1e3, r.queue_sizes()[:10]))
    r.close()
