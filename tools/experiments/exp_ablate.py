import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import dxrs_amd_loader, dxrs_amd
from dxrs_amd.binding import HipLib
host = dxrs_amd.load_host()
s,m,sd = host.scene(0,0)
W,H=1920,1080
cam = host.camera(W,H)
def run(lib, env, reps=30):
    for k,v in env.items(): os.environ[k]=str(v)
    r = dxrs_amd.Renderer(lib=lib)
    r.set_scene(s,m,sd); r.set_camera(cam)
    gs = dxrs_amd.types.graphics_settings(W,H,bounces=8)
    r.set_constants(gs); r.set_profiling(True)
    for _ in range(3): r.render()
    tt=[];tr=[];tl=[]
    for _ in range(reps):
        img, st = r.render(); tt.append(st.ms_total); tr.append(st.ms_traverse); tl.append(st.ms_tail)
    r.close()
    for k in env: os.environ.pop(k)
    return "total %.3f wf %.3f loop %.3f rays %d"%(np.median(tt), np.median(tr), np.median(tl), st.rays)
print("full       ", run(None, {}))
for a,name in ((1,"no BSDF    "),(2,"no traverse"),(3,"neither    ")):
    lib = HipLib(os.path.join(os.getcwd(),"tools","experiments","libpt_ablate%d.so"%a))
    print(name, run(lib, {}))
