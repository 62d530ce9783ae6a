"""C2 primary pass alone (Bounces = int(argv[1]), default 0; looping pass takes the rest): a target for rocprofv3 --pmc."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H = 1920, 1080
os.environ["PT_TAIL_AFTER"] = "0"
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
buf = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
r = dxrs_amd.Renderer(stream=ts.cuda_stream, frames_in_flight=1)
r.set_scene(s, m, sd); r.set_camera(host.camera(W, H))
r.set_constants(dxrs_amd.types.graphics_settings(W, H, bounces=int(sys.argv[1]) if len(sys.argv) > 1 else 0))
for k in range(20): r.render_device(buf.data_ptr())
r.synchronize(); r.close()
