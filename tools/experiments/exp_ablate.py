"""Where the shading time goes: C2 frame timing with parts of shade_step compiled out (temporary -DPT_ABLATE=1|2|3 builds of
csrc/pt_kernels.hip placed in tools/experiments/libpt_ablate<N>.so; 1 = no lobe weights (EnvironmentTerm_Rtg), 2 = no
pdf / eval, 3 = no BSDF sampling).  Images are wrong by construction; only the times mean something."""
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
from dxrs_amd.binding import HipLib
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H = 1920, 1080
cam = host.camera(W, H)


def run(lib, reps=30):
    r = dxrs_amd.Renderer(lib=lib)
    r.set_scene(s, m, sd); r.set_camera(cam)
    gs = dxrs_amd.types.graphics_settings(W, H, bounces=8)
    r.set_constants(gs); r.set_profiling(True)
    for _ in range(3): r.render()
    tt, tr, tl = [], [], []
    for _ in range(reps):
        img, st = r.render(); tt.append(st.ms_total); tr.append(st.ms_traverse); tl.append(st.ms_tail)
    r.close()
    return "total %.3f ms  compacting passes %.3f  loop %.3f  rays %d" % (np.median(tt), np.median(tr), np.median(tl), st.rays)


print("full            ", run(None))
for a, name in ((1, "no lobe weights "), (2, "no pdf / eval   "), (3, "no BSDF sampling")):
    path = os.path.join(os.getcwd(), "tools", "experiments", "libpt_ablate%d.so" % a)
    if os.path.exists(path):
        print(name, run(HipLib(path)))
