import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, time
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s,m,sd = host.scene(0,0)
W,H=1920,1080
cam = host.camera(W,H)
def run(env, bounces=8, reps=30):
    for k,v in env.items(): os.environ[k]=str(v)
    r = dxrs_amd.Renderer()
    r.set_scene(s,m,sd); r.set_camera(cam)
    gs = dxrs_amd.types.graphics_settings(W,H,bounces=bounces)
    r.set_constants(gs)
    r.set_profiling(True)
    for _ in range(3): r.render()
    tt=[];tr=[];sh=[];tl=[]
    for _ in range(reps):
        img, st = r.render()
        tt.append(st.ms_total); tr.append(st.ms_traverse); sh.append(st.ms_shade); tl.append(st.ms_tail)
    r.close()
    for k in env: os.environ.pop(k)
    return np.median(tt), np.median(tr), np.median(sh), np.median(tl), st.rays
os.environ["PT_DEBUG_COUNTS"]="1"
print(run({"PT_TAIL_AFTER":8}, reps=1))
os.environ.pop("PT_DEBUG_COUNTS")
for b in (0,1,2,3,4,6,8):
    print("bounces",b, run({"PT_TAIL_AFTER":99}, bounces=b))
for bpc in (1,2,4,8,16):
    print("trav blocks/cu",bpc, run({"PT_TAIL_AFTER":99,"PT_TRAVERSE_BLOCKS_PER_CU":bpc}))
for bpc in (1,2,4,8,16,32):
    print("shade blocks/cu",bpc, run({"PT_TAIL_AFTER":99,"PT_SHADE_BLOCKS_PER_CU":bpc}))
for bpc in (1,2,4,8):
    print("tail blocks/cu",bpc, run({"PT_TAIL_AFTER":1,"PT_TAIL_BLOCKS_PER_CU":bpc}))
