import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
from test_gpu_fuzz import random_scene
host = dxrs_amd.load_host(); oracle = load_oracle()
seed=80
rng = np.random.default_rng(1000 + seed)
n = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 200]))
spheres, materials = random_scene(dxrs_amd, rng, n)
sd = host.scene(1)[2]
if seed % 3 == 0:
    sd.EnvironmentLightColor[0], sd.EnvironmentLightColor[1], sd.EnvironmentLightColor[2], sd.EnvironmentLightColor[3] = 0.7, 0.8, 1.1, 1.0
w, h = int(rng.choice([48, 64, 81])), int(rng.choice([40, 48, 57]))
bounces, spp = int(rng.choice([0, 1, 3, 6, 12])), int(rng.choice([1, 2, 5]))
pos = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -12.0) if seed % 4 else (0.1, 0.2, 0.3)
cam = host.camera(w, h, position=pos, look_at=(0.0, 0.0, 0.0) if seed % 2 else None, jitter_index=seed)
gs = dxrs_amd.types.graphics_settings(w, h, frame_index=seed * 7919, bounces=bounces, spp=spp, rr=bool(seed % 5))
print("n",n,"w,h",w,h,"bounces",bounces,"spp",spp,"pos",pos, "rr", bool(seed%5))
ref, ost = oracle.render(spheres, materials, sd, cam, gs, threads=8)
for flags in (0, 8, 1, 9):
    r = dxrs_amd.Renderer(flags=flags)
    r.set_scene(spheres, materials, sd); r.set_camera(cam); r.set_constants(gs)
    img, st = r.render()
    bad = (img.view(np.uint32)[...,:3]!=ref.view(np.uint32)[...,:3]).any(-1)
    print("flags",flags,"rays",st.rays,ost.rays,"bad px",bad.sum(), np.argwhere(bad)[:4].tolist())
    if flags==0 and bad.any():
        y,x = np.argwhere(bad)[0]
        print(" gpu",img[y,x],"ref",ref[y,x])
        ev = oracle.trace_pixel(spheres,materials,sd,cam,gs,int(x),int(y))
        for e in ev[:12]: print("     s%d b%d id=%d t=%.6g lobe=%d flag=%d T=(%.3g %.3g %.3g)"%(e[0],e[1],e[2:3].view(np.uint32)[0] if e[2:3].view(np.uint32)[0]!=0xFFFFFFFF else -1,e[3],e[14],e[15],e[10],e[11],e[12]))
        # closest hit check for the primary ray via trace hooks
    r.close()
