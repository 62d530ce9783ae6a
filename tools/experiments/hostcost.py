import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import numpy as np
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s,m,sd = host.scene(0,0)
for (W,H) in ((256,144),(640,384),(1920,1080)):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    r = dxrs_amd.Renderer(stream=ts.cuda_stream, flags=16)
    r.set_scene(s,m,sd)
    gs = dxrs_amd.types.graphics_settings(W,H)
    cams=[host.camera(W,H,jitter_index=k) for k in range(8)]
    bufs=[torch.empty((H*W,4),dtype=torch.float32,device="cuda") for _ in range(2)]
    def step(k):
        gs.FrameIndex=k; r.set_camera(cams[k%8]); r.set_constants(gs); r.render_device(bufs[k%2].data_ptr())
    for k in range(50): step(k)
    torch.cuda.synchronize()
    N=500
    t0=time.perf_counter()
    for k in range(N): step(k)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(W,H,"host enqueue %.1f us/step, total %.1f us/step"%((t1-t0)/N*1e6,(t2-t0)/N*1e6))
    r.close()
