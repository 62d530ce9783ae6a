import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
host = dxrs_amd.load_host(); oracle = load_oracle()
sph = np.zeros(1, dtype=dxrs_amd.SPHERE_DTYPE); sph["r"] = 2.0; sph["cz"] = 1.0
m = dxrs_amd.types.default_material(1); m["BaseColor"] = (0.8, 0.3, 0.2, 1); m["Transmission"] = 1; m["Roughness"] = 0.1
sd = host.scene(1)[2]
for spp,b in ((1,6),(3,6),(2,1)):
    gs = dxrs_amd.types.graphics_settings(96, 64, bounces=b, spp=spp)
    cam = host.camera(96, 64, position=(0, 0, -6))
    ref, ost = oracle.render(sph, m, sd, cam, gs, threads=4)
    for flags in (0,8):
        r = dxrs_amd.Renderer(flags=flags)
        r.set_scene(sph, m, sd); r.set_camera(cam); r.set_constants(gs)
        for rep in range(2):
            img, st = r.render()
            print("spp",spp,"b",b,"flags",flags,"rep",rep,"rays",st.rays,"oracle",ost.rays,"bad",(img.view(np.uint32)[...,:3]!=ref.view(np.uint32)[...,:3]).any(-1).sum())
        r.close()
