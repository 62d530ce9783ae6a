import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
host = dxrs_amd.load_host(); oracle = load_oracle()
s,m,sd = host.scene(1,0)
for spp,b,env in ((2,2,{}),(2,2,{"PT_LOOP_USE_TAIL":"1"})):
    for k,v in env.items(): os.environ[k]=v
    gs = dxrs_amd.types.graphics_settings(128, 96, bounces=b, spp=spp)
    cam = host.camera(128, 96)
    ref, ost = oracle.render(s, m, sd, cam, gs, threads=4)
    r = dxrs_amd.Renderer()
    r.set_scene(s, m, sd); r.set_camera(cam); r.set_constants(gs)
    for rep in range(2):
        img, st = r.render()
        bad = (img.view(np.uint32)[...,:3]!=ref.view(np.uint32)[...,:3]).any(-1)
        print("spp",spp,"b",b,env,"rays",st.rays,"oracle",ost.rays,"bad",bad.sum())
    r.close()
    for k in env: os.environ.pop(k)
