import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import ctypes as C, numpy as np
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle, declare_leaf_api
from test_gpu_accel import make_rays, candidates
host = dxrs_amd.load_host(); oracle = load_oracle(); lib = oracle.lib
dev = C.CDLL(os.path.join(os.getcwd(),'tests','hostshim','libdevmath_host.so')); declare_leaf_api(dev,'dev_')
spheres, materials, sd = host.scene(2, seed=1, count=3000)
r = dxrs_amd.Renderer()
r.set_scene(spheres, materials, sd)
o, d = make_rays(spheres, 200000, seed=len(spheres))
sub = slice(0,2000)
t_bvh, id_bvh = r.trace_rays(o[sub], d[sub], tmin=0.0, use_bvh=True)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
nbad=0
for i in range(2000):
    oi, di = np.ascontiguousarray(o[i]), np.ascontiguousarray(d[i])
    best, best_id = np.float32(np.inf), 0xFFFFFFFF
    tt = C.c_float()
    for sid in range(len(spheres)):
        if lib.oracle_intersect_sphere(fp(oi), fp(di), C.c_float(0.0), C.c_float(best), spheres[sid:sid+1].ctypes.data, C.byref(tt)):
            best, best_id = np.float32(tt.value), sid
    if best_id != id_bvh[i] or (best_id != 0xFFFFFFFF and best != t_bvh[i]):
        nbad+=1
        g = int(id_bvh[i])
        print("ray",i,"o",oi,"d",di,"|d|^2",float((di.astype(np.float64)**2).sum()))
        print("  gpu id",g,"t",t_bvh[i],"oracle id",best_id,"t",best)
        if g != 0xFFFFFFFF:
            s = spheres[g:g+1]; print("  sphere", s)
            h1 = lib.oracle_intersect_sphere(fp(oi), fp(di), C.c_float(0.0), C.c_float(np.inf), s.ctypes.data, C.byref(tt)); t1=tt.value
            h2 = dev.dev_intersect_sphere(fp(oi), fp(di), C.c_float(0.0), C.c_float(np.inf), s.ctypes.data, C.byref(tt)); t2=tt.value
            print("  oracle on that sphere:",h1,t1," dev-host:",h2,t2, "prefilter", candidates(spheres,oi,di)[g])
        if nbad>5: break
print("bad",nbad)
