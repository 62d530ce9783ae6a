import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
import dxrs_amd_loader, dxrs_amd
from oracle.binding import load_oracle
host = dxrs_amd.load_host(); o = load_oracle()
for name,(kind,count) in {"demo":(0,0),"proc1M":(2,1<<20)}.items():
    s,m,sd = host.scene(kind, seed=1 if kind==2 else 0, count=count)
    W,H=1920,1080
    cam = host.camera(W,H)
    r = dxrs_amd.Renderer(); info = r.set_scene(s,m,sd)
    # primary rays on a coarse pixel grid
    oo=[];dd=[]
    for py in range(4,H,16):
        for px in range(4,W,16):
            a=np.zeros(3,np.float32); b=np.zeros(3,np.float32); t0=C.c_float(); t1=C.c_float()
            o.lib.oracle_primary_ray(C.addressof(cam),px,py,W,H,a.ctypes.data_as(C.POINTER(C.c_float)),b.ctypes.data_as(C.POINTER(C.c_float)),C.byref(t0),C.byref(t1))
            oo.append(a);dd.append(b)
    oo=np.array(oo);dd=np.array(dd)
    t,ids,v = r.trace_rays_stats(oo,dd)
    print(name,"depth",info.depth,"primary rays",len(oo),"hit frac %.2f"%(ids!=0xFFFFFFFF).mean(),"nodes mean %.1f p50 %d p99 %d max %d | spheres mean %.2f max %d"%(v[:,0].mean(),np.median(v[:,0]),np.percentile(v[:,0],99),v[:,0].max(),v[:,1].mean(),v[:,1].max()))
    # per-wave max (8x8 blocks): approximate by grouping 64 consecutive rays (coarse grid rows)
    # secondary: random directions from hit points
    hit = ids!=0xFFFFFFFF
    P = oo[hit]+dd[hit]*t[hit,None]
    rng=np.random.default_rng(0); L=rng.normal(size=P.shape); L/=np.linalg.norm(L,axis=1,keepdims=True); L[:,1]=np.abs(L[:,1])
    t2,ids2,v2 = r.trace_rays_stats(P+L*1e-3, L.astype(np.float32))
    print("   secondary (random upward dirs from hits): nodes mean %.1f p99 %d max %d | spheres mean %.2f"%(v2[:,0].mean(),np.percentile(v2[:,0],99),v2[:,0].max(),v2[:,1].mean()))
    r.close()
