import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings
w, h, lanes = 1920, 1080, 3
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]
def run(tag, rebuild, textured):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    r = dxrs_amd.Renderer(device=0, stream=ts.cuda_stream, frames_in_flight=lanes)
    a = r.set_scene(spheres, materials, sd)
    if rebuild: a = r.build_accel()
    if textured: r.set_textures(host.demo_textures(0, 0.0, textured=True))
    gs = graphics_settings(w, h, frame_index=0, bounces=8, spp=1)
    bufs = [torch.empty((h * w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
    def frame(k, stats=False):
        gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs); return r.render_device(bufs[k % lanes].data_ptr(), want_stats=stats)
    for k in range(30): frame(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(300): frame(30 + k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e3
    st = frame(400, stats=True)
    print(f"{tag:50s} {dt:.4f} ms/frame; nodes {a.node_count} depth {a.depth} lds {a.lds_resident}; stats: beams_used {getattr(st, 'beams_used', None)} rays {st.rays} ms {st.ms_total if hasattr(st,'ms_total') else ''}")
    r.close()
run("untextured, set_scene only", False, False)
run("untextured, set_scene + build_accel", True, False)
run("textured, set_scene only", False, True)
run("textured, set_scene + build_accel", True, True)
