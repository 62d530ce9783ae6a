"""ctypes access to libpt_host.so: the C++ host mirror (scene generators, camera controller, Halton jitter)."""
import ctypes as C
import math
import os

import numpy as np

from .abi_types import MATERIAL_DTYPE, SPHERE_DTYPE, PtCamera, PtSceneData

_PKG = os.path.dirname(os.path.abspath(__file__))

SCENE_DEMO, SCENE_SMALL, SCENE_PROCEDURAL = 0, 1, 2


class HostLib:
    def __init__(self, path=None):
        path = path or os.path.join(_PKG, "libpt_host.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        self.lib = C.CDLL(path)
        self.lib.pth_scene.restype = C.c_int
        self.lib.pth_scene.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(PtSceneData)]
        self.lib.pth_camera.restype = None
        self.lib.pth_camera.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(PtCamera)]
        self.lib.pth_scene_at_time.restype = C.c_int
        self.lib.pth_scene_at_time.argtypes = [C.c_uint32, C.c_double, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        self.lib.pth_halton.restype = C.c_float
        self.lib.pth_halton.argtypes = [C.c_uint32, C.c_uint32]
        self.lib.pth_random_floats.restype = None
        self.lib.pth_random_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]

    def scene(self, kind=SCENE_DEMO, seed=0, count=0):
        """-> (spheres[SPHERE_DTYPE], materials[MATERIAL_DTYPE], PtSceneData)"""
        n = C.c_uint32(0)
        sd = PtSceneData()
        rc = self.lib.pth_scene(kind, seed, count, None, None, 0, C.byref(n), C.byref(sd))
        if rc:
            raise ValueError(f"pth_scene size query failed ({rc})")
        spheres = np.zeros(n.value, dtype=SPHERE_DTYPE)
        materials = np.zeros(n.value, dtype=MATERIAL_DTYPE)
        rc = self.lib.pth_scene(kind, seed, count, spheres.ctypes.data, materials.ctypes.data, n.value, C.byref(n), C.byref(sd))
        if rc:
            raise ValueError(f"pth_scene failed ({rc})")
        return spheres, materials, sd

    def scene_at_time(self, seed, time):
        """Spheres of the demo scene at simulation time `time` (closed-form springs + Moon orbit, MyScene::SetTime)."""
        n = C.c_uint32(0)
        self.lib.pth_scene_at_time(seed, time, None, 0, C.byref(n))
        spheres = np.zeros(n.value, dtype=SPHERE_DTYPE)
        rc = self.lib.pth_scene_at_time(seed, time, spheres.ctypes.data, n.value, C.byref(n))
        if rc:
            raise ValueError(f"pth_scene_at_time failed ({rc})")
        return spheres

    def demo_textures(self, seed=0, time=0.0, textured=True, environment_map=False, return_scene_data=False, texture_dir=None):
        """TextureSet of the demo scene's textured objects (Alien-Metal, Moon, Earth: Source/MyScene.ixx:161-166, 285-295)
        at simulation time `time`, as the C++ host mirror builds it (MySceneDesc(seed, textured, environmentMap); procedural
        stand-ins for the reference's image files unless a loader is installed on the C++ side).  environment_map adds the
        lat-long environment light (MyScene.ixx:94-95) to the table; return_scene_data -> (TextureSet, PtSceneData), the
        SceneData that names it.  texture_dir: a directory of decoded images (<stem>.ptex, host/Texture.hpp LoadRawTexture -- e.g.
        tests/golden/textures, the reference's own assets) to use instead of the stand-ins."""
        flags = (1 if textured else 0) | (2 if environment_map else 0)
        saved = os.environ.get("PT_TEXTURE_DIR")
        if texture_dir is not None:
            os.environ["PT_TEXTURE_DIR"] = str(texture_dir)  # read by DefaultTextureLoader inside the calls below
        elif saved is not None:
            del os.environ["PT_TEXTURE_DIR"]
        try:
            return self._demo_textures(seed, time, flags, return_scene_data)
        finally:
            if saved is not None:
                os.environ["PT_TEXTURE_DIR"] = saved
            else:
                os.environ.pop("PT_TEXTURE_DIR", None)

    def _demo_textures(self, seed, time, flags, return_scene_data):
        from .abi_types import TEXTURE_MAP_COUNT, TEXTURE_RGBA8_UNORM_SRGB, TEXTURE_RGBA32_FLOAT, PtObjectTextures, PtSceneData
        from .textures import TextureSet
        fn = self.lib.pth_demo_textures_ex
        fn.restype = C.c_int
        fn.argtypes = [C.c_uint32, C.c_double, C.c_uint32] + [C.c_void_p] * 8
        nt, no, nb = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        sd = PtSceneData()
        fn(seed, time, flags, C.byref(nt), C.byref(no), C.byref(nb), None, None, None, None, None)
        info = np.zeros((nt.value, 4), dtype=np.uint32)
        pixels = np.zeros(nb.value, dtype=np.uint8)
        obj = (PtObjectTextures * no.value)()
        rot = np.zeros((no.value, 4), dtype=np.float32)
        rc = fn(seed, time, flags, C.byref(nt), C.byref(no), C.byref(nb), info.ctypes.data, pixels.ctypes.data, C.addressof(obj), rot.ctypes.data, C.addressof(sd))
        if rc:
            raise ValueError(f"pth_demo_textures_ex failed ({rc})")
        ts = TextureSet(no.value)
        for (w, h, fmt, off) in info:
            w, h, off = int(w), int(h), int(off)
            if fmt == TEXTURE_RGBA32_FLOAT:
                ts.add_hdr_image(pixels[off:off + w * h * 16].view(np.float32).reshape(h, w, 4).copy())
            else:
                ts.add_image(pixels[off:off + w * h * 4].reshape(h, w, 4).copy(), srgb=fmt == TEXTURE_RGBA8_UNORM_SRGB)
        for i in range(no.value):
            for k in range(TEXTURE_MAP_COUNT):
                ts.maps[i, k] = obj[i].Maps[k].Descriptor
        ts.rotations[:] = rot
        return (ts, sd) if return_scene_data else ts

    def camera(self, width, height, position=(0.0, 0.0, -15.0), look_at=None, hfov=math.pi / 2, jitter=True, jitter_index=0, jitter_count=8):
        """Demo camera (MyScene.ixx:90; HFOV 90 deg, MyAppData.h:177); jitter = Halton2D(index + 1) - 0.5 cycling mod 8."""
        cam = PtCamera()
        pos = (C.c_float * 3)(*position)
        la = (C.c_float * 3)(*look_at) if look_at is not None else None
        self.lib.pth_camera(pos, la, hfov, width, height, 1 if jitter else 0, jitter_index, jitter_count, C.byref(cam))
        return cam

    def halton(self, index, base):
        return float(self.lib.pth_halton(index, base))

    def random_floats(self, seed, n):
        out = np.zeros(n, dtype=np.float32)
        self.lib.pth_random_floats(seed, n, out.ctypes.data)
        return out


_host = None


def load_host():
    global _host
    if _host is None:
        _host = HostLib()
    return _host
