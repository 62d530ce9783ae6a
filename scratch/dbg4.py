import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
host = dxrs_amd.load_host(); oracle = load_oracle()
s,m,sd = host.scene(1,0)
spp,b=2,2
gs = dxrs_amd.types.graphics_settings(128, 96, bounces=b, spp=spp)
cam = host.camera(128, 96)
ref, ost = oracle.render(s, m, sd, cam, gs, threads=4)
# per-sample oracle references: sample 0 only (spp=1)
gs1 = dxrs_amd.types.graphics_settings(128, 96, bounces=b, spp=1)
ref1,_ = oracle.render(s, m, sd, cam, gs1, threads=4)
r = dxrs_amd.Renderer()
r.set_scene(s, m, sd); r.set_camera(cam); r.set_constants(gs)
img, st = r.render()
bad = (img.view(np.uint32)[...,:3]!=ref.view(np.uint32)[...,:3]).any(-1)
for (y,x) in np.argwhere(bad)[:8]:
    print(y,x,"gpu",img[y,x,:3],"oracle",ref[y,x,:3],"oracle spp1",ref1[y,x,:3], "2*ref - ref1 (sample1 alone)", 2*ref[y,x,:3]-ref1[y,x,:3])
    ev = oracle.trace_pixel(s,m,sd,cam,gs,int(x),int(y))
    for e in ev: print("     s%d b%d id=%d lobe=%d flag=%d T=(%.3g %.3g %.3g)"%(e[0],e[1],e[2:3].view(np.uint32)[0] if e[2:3].view(np.uint32)[0]!=0xFFFFFFFF else -1,e[14],e[15],e[10],e[11],e[12]))
