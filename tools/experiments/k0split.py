"""What the primary pass spends its time on: exclusive duration of the first launch (bounce_kernel<primary>) of a C2 frame
for Bounces = 8 (trace + full shade + compaction) and Bounces = 0 (trace + hit frame + emission + pixel store only)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H = 1920, 1080
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
buf = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
for bounces in (8, 0):
    os.environ["PT_TAIL_AFTER"] = "0"  # after the primary pass the looping kernel takes everything: launch 1 = primary pass alone
    r = dxrs_amd.Renderer(stream=ts.cuda_stream, frames_in_flight=1)
    r.set_scene(s, m, sd); r.set_camera(host.camera(W, H))
    gs = dxrs_amd.types.graphics_settings(W, H, bounces=bounces); r.set_constants(gs)
    r.set_profiling(True)
    for k in range(5): r.render_device(buf.data_ptr())
    r.synchronize(); r.profile(reset=True)
    N = 50
    for k in range(N): r.render_device(buf.data_ptr())
    r.synchronize()
    p = r.profile(reset=True)
    print("Bounces %d: primary pass %.1f us   (looping pass %.1f us)" % (bounces, p.ms_traverse / max(p.traverse_launches, 1) * 1e3, p.ms_tail / max(p.tail_launches, 1) * 1e3))
    r.close()
