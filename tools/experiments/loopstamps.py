"""Where a wave of the looping pass spends its cycles (temporary -DPT_STAMPS build of pt_kernels.hip, hooks not kept in the tree):
s_memtime stamps around the cursor fetch, the path load, every trace and every shade step, summed over the waves of a launch.
usage: PT_HIP_LIB=tools/experiments/libpt_stamps.so python tools/experiments/loopstamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dxrs_amd_loader  # noqa
import dxrs_amd
from dxrs_amd.types import graphics_settings

lib = ctypes.CDLL(os.environ["PT_HIP_LIB"])
w, h = 1920, 1080
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
r = dxrs_amd.Renderer(device=0, frames_in_flight=1)
r.set_scene(spheres, materials, sd)
buf = torch.empty((h * w, 4), dtype=torch.float32, device="cuda")
st = (ctypes.c_ulonglong * 16)()
for bounces in (2, 3, 8):
    gs = graphics_settings(w, h, frame_index=0, bounces=bounces, spp=1)
    for k in range(6):
        gs.FrameIndex = k; r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs); r.render_device(buf.data_ptr())
    torch.cuda.synchronize()
    lib.pt_debug_stamps(st, 1)
    r.set_profiling(True)
    n = 10
    for k in range(n):
        gs.FrameIndex = k; r.set_camera(host.camera(w, h, jitter_index=k)); r.set_constants(gs); r.render_device(buf.data_ptr())
    p = r.profile(reset=True)
    r.set_profiling(False)
    torch.cuda.synchronize()
    lib.pt_debug_stamps(st, 1)
    v = [int(x) for x in st]
    waves, chunks, iters = v[5] / n, v[6] / n, v[7] / n
    print(f"bounces {bounces}: loop {p.ms_tail / max(p.tail_launches, 1) * 1e3:6.1f} us  queue {r.queue_sizes()[:3]}")
    print(f"   waves/launch {waves:.0f}  chunks {chunks:.0f}  wave-level bounce iterations {iters:.0f} ({iters / max(chunks, 1):.2f} per chunk; max per wave {v[11]}, max chunks per wave {v[12]})")
    print(f"   per wave (cycles): prolog {v[0] / v[5]:.0f} (max {v[13]})  lifetime mean {v[9] / v[5]:.0f} max {v[8]} (max wall clock {v[10]} ticks of 100 MHz)")
    print(f"   per chunk (cycles): cursor {v[1] / max(v[6], 1):.0f}  path load {v[2] / max(v[6], 1):.0f};  per iteration: trace {v[3] / max(v[7], 1):.0f}  shade {v[4] / max(v[7], 1):.0f}")
r.close()
