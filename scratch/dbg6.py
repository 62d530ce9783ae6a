import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s,m,sd = host.scene(0,0)
w,h,n=640,360,8
cams=[host.camera(w,h,jitter_index=k) for k in range(8)]
gs = dxrs_amd.types.graphics_settings(w,h,bounces=8,spp=1)
r = dxrs_amd.Renderer(); r.set_scene(s,m,sd)
ref=[]
for k in range(n):
    gs.FrameIndex=k; r.set_camera(cams[k%8]); r.set_constants(gs); img,st=r.render(); ref.append(img)
stream = torch.cuda.current_stream().cuda_stream
for sync_each in (True, False):
    r2 = dxrs_amd.Renderer(stream=stream, flags=16); r2.set_scene(s,m,sd)
    bufs=[torch.empty((h,w,4),dtype=torch.float32,device="cuda") for _ in range(2)]
    keep=[]
    for k in range(n):
        gs.FrameIndex=k; r2.set_camera(cams[k%8]); r2.set_constants(gs)
        r2.render_device(bufs[k%2].data_ptr())
        if sync_each: r2.synchronize()
        keep.append(bufs[k%2].clone())
    torch.cuda.synchronize()
    for k in range(n):
        a=keep[k].cpu().numpy(); bad=(a.view(np.uint32)!=ref[k].view(np.uint32)).any(-1)
        print("sync_each",sync_each,"frame",k,"bad px",bad.sum(), np.argwhere(bad)[:3].tolist())
    r2.close()
