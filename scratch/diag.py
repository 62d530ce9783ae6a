import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
host = dxrs_amd.load_host(); o = load_oracle()
s,m,sd = host.scene(1,0)
gs = dxrs_amd.types.graphics_settings(256,256,bounces=4)
cam = host.camera(256,256)
ref, ost = o.render(s,m,sd,cam,gs,threads=8)
for ta in ("8","3","2","1","0"):
    os.environ["PT_TAIL_AFTER"]=ta
    r = dxrs_amd.Renderer()
    r.set_scene(s,m,sd); r.set_camera(cam); r.set_constants(gs)
    for rep in range(2):
        img, st = r.render()
        bad = (img.view(np.uint32)[...,:3]!=ref.view(np.uint32)[...,:3]).any(-1).sum()
        print("tail_after",ta,"rep",rep,"rays",st.rays,"oracle",ost.rays,"bad px",bad)
    r.close()
