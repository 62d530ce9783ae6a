import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, time
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s,m,sd = host.scene(0,0)
W,H=1920,1080
cam = host.camera(W,H)
def run(env, bounces=8, reps=30, flags=0):
    for k,v in env.items(): os.environ[k]=str(v)
    r = dxrs_amd.Renderer(flags=flags)
    r.set_scene(s,m,sd); r.set_camera(cam)
    gs = dxrs_amd.types.graphics_settings(W,H,bounces=bounces)
    r.set_constants(gs)
    r.set_profiling(True)
    for _ in range(3): r.render()
    tt=[];tr=[];sh=[];tl=[]
    for _ in range(reps):
        img, st = r.render()
        tt.append(st.ms_total); tr.append(st.ms_traverse); sh.append(st.ms_shade); tl.append(st.ms_tail)
    qs = r.queue_sizes()
    r.close()
    for k in env: os.environ.pop(k)
    return "total %.3f trav/fused %.3f shade %.3f loop %.3f"%(np.median(tt), np.median(tr), np.median(sh), np.median(tl)), st.rays, qs[:6]
for ta in (0,1,2,3):
    print("fused tail_after",ta, run({"PT_TAIL_AFTER":ta}))
for th in (256,512):
    print("fused threads",th, run({"PT_FUSED_THREADS":th}))
for ta in (1,2):
    print("split tail_after",ta, run({"PT_TAIL_AFTER":ta}, flags=8))
