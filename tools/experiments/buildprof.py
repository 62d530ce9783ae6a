"""LBVH build at 2^20 spheres: run under `rocprofv3 --kernel-trace --stats` to see the per-kernel split."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(dxrs_amd.host.SCENE_PROCEDURAL, seed=1, count=int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20)
r = dxrs_amd.Renderer()
r.set_scene(s, m, sd)
for _ in range(5):
    a = r.build_accel()
print("build_ms", a.build_ms, "depth", a.depth)
r.close()
