"""Host and GPU cost per frame of the tiled path for one rank's share (rehearsed with world = 1 over NCCL):
where does the time go when the share is small (1/8 of a 1080p frame = 640x384-equivalent)?"""
import sys, os, time; sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H = 640, 384
for lanes in ([int(a) for a in sys.argv[1:]] or [1, 4, 8]):
    ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
    r = dxrs_amd.Renderer(stream=ts.cuda_stream, frames_in_flight=lanes)
    r.set_scene(s, m, sd); r.set_partition(0, 1)
    gs = dxrs_amd.types.graphics_settings(W, H); r.set_constants(gs)
    cams = [host.camera(W, H, jitter_index=k) for k in range(8)]
    mt = r.tiles_count(0)
    packeds = [torch.zeros((mt * 1024, 4), dtype=torch.float32, device=dev) for _ in range(lanes)]
    gathered = torch.empty((1, mt * 1024, 4), dtype=torch.float32, device=dev); gl = list(gathered.unbind(0))
    frame = torch.empty((H * W, 4), dtype=torch.float32, device=dev)
    acc = [0.0] * 4
    def step(k, mode):
        t0 = time.perf_counter()
        gs.FrameIndex = k; r.set_camera(cams[k % 8]); r.set_constants(gs)
        t1 = time.perf_counter()
        r.render_tiles(packeds[k % lanes].data_ptr())
        t2 = time.perf_counter()
        if mode >= 1: dist.gather(packeds[k % lanes], gl, dst=0)
        t3 = time.perf_counter()
        if mode >= 2: r.unpack_tiles(gathered.data_ptr(), mt, frame.data_ptr())
        t4 = time.perf_counter()
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3
    for mode in (0, 1, 2):
        for k in range(50): step(k, mode)
        torch.cuda.synchronize()
        acc[:] = [0.0] * 4
        N = 500
        t0 = time.perf_counter()
        for k in range(N): step(k, mode)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"lanes {lanes} mode {mode} (0 render, 1 +gather, 2 +unpack): host {(t1 - t0) / N * 1e6:.1f} us/frame, total {(t2 - t0) / N * 1e6:.1f} us/frame; "
              f"host split set {acc[0] / N * 1e6:.1f} render {acc[1] / N * 1e6:.1f} gather {acc[2] / N * 1e6:.1f} unpack {acc[3] / N * 1e6:.1f}", flush=True)
    r.close()
dist.destroy_process_group()
