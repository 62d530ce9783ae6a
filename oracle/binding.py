"""ctypes access to the CPU ORACLE (oracle/libpt_oracle.so).  TEST INFRASTRUCTURE ONLY: import from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product package."""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


class OracleStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64)]


class OracleBsdfOut(C.Structure):
    _fields_ = [("lobe", C.c_int), ("valid", C.c_int), ("L", C.c_float * 3), ("pdf", C.c_float), ("f", C.c_float * 3), ("weights", C.c_float * 3)]


class OracleTextures(C.Structure):
    _fields_ = [("textures", C.c_void_p), ("n_textures", C.c_uint32), ("object_textures", C.c_void_p), ("rotations", C.c_void_p)]


class _Rect(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


def declare_leaf_api(lib, prefix):
    """Declare the leaf-function signatures shared by the oracle (prefix 'oracle_') and the host-compiled device
    headers test shim (prefix 'dev_')."""
    f, u32, vp = C.c_float, C.c_uint32, C.c_void_p
    pf = C.POINTER(C.c_float)

    def fn(name, res, args):
        func = getattr(lib, prefix + name)
        func.restype = res
        func.argtypes = args
        return func

    fn("hash", u32, [u32])
    fn("rng_init", u32, [u32, u32, u32])
    fn("rng_next", u32, [C.POINTER(u32)])
    fn("rng_float", f, [C.POINTER(u32)])
    fn("sincos_2pi", None, [f, pf, pf])
    fn("log2", f, [f])
    fn("exp2", f, [f])
    fn("pow", f, [f, f])
    fn("from_srgb", f, [f])
    fn("get_basis", None, [pf, pf, pf])
    fn("cosine_ray", None, [pf, pf])
    fn("vndf_ray", None, [pf, f, pf, pf])
    fn("vndf_pdf", f, [pf, f, f])
    fn("distribution_term", f, [f, f])
    fn("geometry_term_mod", f, [f, f, f])
    fn("fresnel_dielectric", f, [f, f])
    fn("diffuse_term", f, [f, f, f, f])
    fn("environment_term_rtg", None, [pf, f, f, pf])
    fn("sky", None, [vp, pf, pf])
    fn("intersect_sphere", C.c_int, [pf, pf, f, f, vp, pf])
    fn("hit_frame", None, [pf, pf, f, vp, pf, pf, pf, C.POINTER(C.c_int)])
    fn("spawn_origin", None, [pf, pf, f, pf, pf])
    fn("primary_ray", None, [vp, u32, u32, u32, u32, pf, pf, pf, pf])
    fn("bsdf_step", None, [vp, C.c_int, pf, pf, pf, C.POINTER(OracleBsdfOut)])
    fn("sample_sphere_cone", C.c_int, [pf, pf, f, f, f, pf, pf])
    fn("atan2", f, [f, f])
    fn("sphere_uv", None, [pf, pf])
    fn("latlong_uv", None, [pf, pf])
    fn("cube_face_uv", C.c_uint32, [pf, pf])
    fn("sphere_tangent", None, [pf, pf])
    fn("quat_rotate", None, [pf, pf, pf])
    fn("perturb_normal", None, [pf, pf, f, f, pf])
    fn("tonemap_pixel", u32, [pf, vp])
    fn("accumulate", None, [vp, vp, u32, u32])


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(_DIR, "libpt_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run `make -C oracle`")
        self.lib = lib = C.CDLL(path)
        vp, u32 = C.c_void_p, C.c_uint32
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [vp, vp, u32, vp, vp, vp, C.POINTER(_Rect), u32, vp, C.POINTER(OracleStats), C.c_int]
        lib.oracle_trace_pixel.restype = C.c_int
        lib.oracle_trace_pixel.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, vp, u32, C.POINTER(u32)]
        lib.oracle_halton.restype = C.c_float
        lib.oracle_halton.argtypes = [u32, u32]
        lib.oracle_render_textured.restype = C.c_int
        lib.oracle_render_textured.argtypes = [vp, vp, u32, vp, vp, vp, C.POINTER(_Rect), u32, vp, C.POINTER(OracleStats), C.c_int, vp]
        lib.oracle_sample_texture.restype = None
        lib.oracle_sample_texture.argtypes = [vp, u32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.oracle_closest_hit.restype = C.c_int
        lib.oracle_closest_hit.argtypes = [vp, u32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int, C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(u32)]
        lib.oracle_closest_hit_alpha.restype = C.c_int
        lib.oracle_closest_hit_alpha.argtypes = [vp, vp, u32, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(u32)]
        lib.oracle_free_bvh.restype = None
        lib.oracle_free_bvh.argtypes = [vp]
        lib.oracle_tonemap.restype = None
        lib.oracle_tonemap.argtypes = [vp, u32, vp, vp]
        declare_leaf_api(lib, "oracle_")

    @staticmethod
    def _textures_struct(texture_set):
        tex, n_tex, obj, rot = texture_set.as_ctypes()
        t = OracleTextures(C.cast(tex, C.c_void_p), n_tex, C.cast(obj, C.c_void_p), rot.ctypes.data)
        return t, (tex, obj, rot)

    def sample_texture(self, texture_set, index, uv):
        t, keep = self._textures_struct(texture_set)
        uvf = (C.c_float * 2)(*[float(x) for x in uv]); out = (C.c_float * 4)()
        self.lib.oracle_sample_texture(C.addressof(t), index, uvf, out)
        return np.array(out[:], dtype=np.float32)

    def render(self, spheres, materials, scene_data, camera, gs, rect=None, row_step=1, threads=1, textures=None):
        """-> (rgba float32 array (h, w, 4), OracleStats).  Rows skipped by row_step are left as NaN.
        textures: a dxrs_amd.textures.TextureSet (row N1) or None."""
        spheres = np.ascontiguousarray(spheres)
        materials = np.ascontiguousarray(materials)
        if rect is None:
            rect = (0, 0, gs.RenderSize[0], gs.RenderSize[1])
        r = _Rect(*rect)
        out = np.full((r.h, r.w, 4), np.nan, dtype=np.float32)
        stats = OracleStats()
        if textures is not None:
            t, keep = self._textures_struct(textures)
            rc = self.lib.oracle_render_textured(spheres.ctypes.data, materials.ctypes.data, len(spheres), C.addressof(scene_data), C.addressof(camera),
                                                 C.addressof(gs), C.byref(r), row_step, out.ctypes.data, C.byref(stats), threads, C.addressof(t))
        else:
            rc = self.lib.oracle_render(spheres.ctypes.data, materials.ctypes.data, len(spheres), C.addressof(scene_data), C.addressof(camera),
                                        C.addressof(gs), C.byref(r), row_step, out.ctypes.data, C.byref(stats), threads)
        if rc:
            raise RuntimeError(f"oracle_render failed ({rc})")
        return out, stats

    def closest_hits(self, spheres, origins, directions, tmin=0.0, tmax=float("inf"), use_bvh=True):
        """closest hit of every ray: (t float32[n], id uint32[n]) through the brute-force loop or the oracle's own BVH"""
        spheres = np.ascontiguousarray(spheres)
        o = np.ascontiguousarray(origins, dtype=np.float32); d = np.ascontiguousarray(directions, dtype=np.float32)
        t = np.zeros(len(o), dtype=np.float32); ids = np.zeros(len(o), dtype=np.uint32)
        cache = C.c_void_p(None)
        pf = C.POINTER(C.c_float)
        for i in range(len(o)):
            tt, ii = C.c_float(), C.c_uint32()
            self.lib.oracle_closest_hit(spheres.ctypes.data, len(spheres), o[i].ctypes.data_as(pf), d[i].ctypes.data_as(pf), tmin, tmax, 1 if use_bvh else 0,
                                        C.byref(cache) if use_bvh else None, C.byref(tt), C.byref(ii))
            t[i], ids[i] = tt.value, ii.value
        if cache.value:
            self.lib.oracle_free_bvh(cache)
        return t, ids

    def closest_hits_alpha(self, spheres, materials, origins, directions, tmin=0.0, tmax=float("inf"), textures=None):
        """closest hit of every ray with alpha-tested hits (spec S10), brute force: (t float32[n], id uint32[n])"""
        spheres = np.ascontiguousarray(spheres); materials = np.ascontiguousarray(materials)
        o = np.ascontiguousarray(origins, dtype=np.float32); d = np.ascontiguousarray(directions, dtype=np.float32)
        t = np.zeros(len(o), dtype=np.float32); ids = np.zeros(len(o), dtype=np.uint32)
        pf = C.POINTER(C.c_float)
        tex, keep = (self._textures_struct(textures) if textures is not None else (None, None))
        for i in range(len(o)):
            tt, ii = C.c_float(), C.c_uint32()
            self.lib.oracle_closest_hit_alpha(spheres.ctypes.data, materials.ctypes.data, len(spheres), C.addressof(tex) if tex is not None else None,
                                              o[i].ctypes.data_as(pf), d[i].ctypes.data_as(pf), tmin, tmax, C.byref(tt), C.byref(ii))
            t[i], ids[i] = tt.value, ii.value
        return t, ids

    def tonemap(self, hdr, params):
        """hdr (..., 4) float32 -> packed uint32 (...): the display transform of row N3"""
        hdr = np.ascontiguousarray(hdr, dtype=np.float32)
        out = np.empty(hdr.shape[:-1], dtype=np.uint32)
        self.lib.oracle_tonemap(hdr.ctypes.data, out.size, C.addressof(params), out.ctypes.data)
        return out

    def accumulate(self, accum, radiance, frames_accumulated):
        """in-place running mean: accum (..., 4) float32 <- radiance"""
        assert accum.flags.c_contiguous and accum.dtype == np.float32 and radiance.shape == accum.shape
        radiance = np.ascontiguousarray(radiance, dtype=np.float32)
        self.lib.oracle_accumulate(accum.ctypes.data, radiance.ctypes.data, accum.size // 4, frames_accumulated)
        return accum

    def trace_pixel(self, spheres, materials, scene_data, camera, gs, px, py, max_events=4096):
        spheres = np.ascontiguousarray(spheres)
        materials = np.ascontiguousarray(materials)
        ev = np.zeros((max_events, 16), dtype=np.float32)
        n = C.c_uint32(0)
        rc = self.lib.oracle_trace_pixel(spheres.ctypes.data, materials.ctypes.data, len(spheres), C.addressof(scene_data), C.addressof(camera),
                                         C.addressof(gs), px, py, ev.ctypes.data, max_events, C.byref(n))
        if rc:
            raise RuntimeError(f"oracle_trace_pixel failed ({rc})")
        return ev[: n.value]


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle
