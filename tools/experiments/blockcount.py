"""Dynamic wave-level execution counts of the shading / traversal blocks of a frame (temporary -DPT_BLOCKCOUNT build of the kernels placed
in tools/experiments/libpt_count.so, hooks not kept in the tree): how often a wave runs the surface-shading code at bounce 0 / 1 / deeper and
with how many active lanes, ditto the miss code, BVH node visits, leaf tests and primary-beam list entries."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PT_HIP_LIB"] = os.path.join(ROOT, "tools", "experiments", "libpt_count.so")
import dxrs_amd_loader  # noqa: F401,E402
import dxrs_amd  # noqa: E402
from dxrs_amd.types import graphics_settings  # noqa: E402

w, h, spp, bounces = [int(x) for x in (sys.argv[1:5] + [1920, 1080, 1, 8][len(sys.argv) - 1:])]
host = dxrs_amd.load_host()
spheres, materials, sd = host.scene(dxrs_amd.host.SCENE_DEMO, seed=0)
r = dxrs_amd.Renderer(device=0, frames_in_flight=1)
r.set_scene(spheres, materials, sd)
r.set_constants(graphics_settings(w, h, bounces=bounces, spp=spp))
r.set_camera(host.camera(w, h))
lib = C.CDLL(os.environ["PT_HIP_LIB"])
out = (C.c_ulonglong * 32)()
names = ["surface b0", "surface b1", "surface b>=2", "miss b0", "miss b1", "miss b>=2", "node visit", "leaf test", "beam list entry"]
for beams in (False, True):
    for k in range(3 if beams else 1):
        r.render()
    lib.pt_debug_blockcount(out, 1)
    img, st = r.render()
    lib.pt_debug_blockcount(out, 1)
    print(f"{w}x{h} {spp} spp, beams_used={st.beams_used}, rays {st.rays}")
    for i, n in enumerate(names):
        wv, ln = out[2 * i], out[2 * i + 1]
        print(f"  {n:16s} wave-executions {wv:10d}  active lanes {ln:12d}  ({ln / max(wv, 1):5.1f} per execution)")
r.close()
