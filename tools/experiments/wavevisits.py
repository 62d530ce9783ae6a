"""Per-wave (8x8 pixel block) LBVH work of the C2 primary rays: a wave executes max-over-lanes node visits / sphere tests,
so the lane mean understates the cost.  Uses pt_trace_rays_stats on every primary ray of the 1080p frame."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dxrs_amd_loader, dxrs_amd
host = dxrs_amd.load_host()
s, m, sd = host.scene(0, 0)
W, H = 1920, 1080
cam = host.camera(W, H, jitter=False)
px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
u, v = (px + 0.5) / W, (py + 0.5) / H
nx, ny = u * 2 - 1, 1 - v * 2
R, U, F = (np.array(x[:3], dtype=np.float32) for x in (cam.RightDirection, cam.UpDirection, cam.ForwardDirection))
d = nx[..., None] * R + ny[..., None] * U + F
d /= np.linalg.norm(d, axis=-1, keepdims=True)
o = np.broadcast_to(np.array(cam.Position[:3], dtype=np.float32), d.shape)
r = dxrs_amd.Renderer(); r.set_scene(s, m, sd)
t, ids, vis = r.trace_rays_stats(o.reshape(-1, 3).copy(), d.reshape(-1, 3).astype(np.float32))
nodes, sph = vis[:, 0].reshape(H, W), vis[:, 1].reshape(H, W)
hb, wb = H // 8, W // 8
blk = lambda a: a[: hb * 8, : wb * 8].reshape(hb, 8, wb, 8).transpose(0, 2, 1, 3).reshape(hb * wb, 64)
bn, bs = blk(nodes), blk(sph)
print("lanes : node visits mean %.1f  sphere tests mean %.2f   hit fraction %.2f" % (nodes.mean(), sph.mean(), (ids != 0xFFFFFFFF).mean()))
print("waves : max node visits mean %.1f (p50 %d, p99 %d)   max sphere tests mean %.2f   => lane utilisation %.0f %% / %.0f %%" % (
    bn.max(1).mean(), np.median(bn.max(1)), np.percentile(bn.max(1), 99), bs.max(1).mean(),
    100 * nodes.mean() / bn.max(1).mean(), 100 * sph.mean() / max(bs.max(1).mean(), 1e-9)))
