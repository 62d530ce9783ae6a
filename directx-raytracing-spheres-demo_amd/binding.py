"""ctypes binding of the C-ABI (include/pt_api.h) exported by libpt_hip.so, plus a thin ``Renderer`` convenience
class shaped like the reference's pass object (Raytracing::SetConstants / Render, Source/Raytracing.ixx:92-112).

No CPU fallback: if the library is absent or there is no GPU, construction raises."""
import ctypes as C
import os

import numpy as np

from .abi_types import (BVH_NODE_DTYPE, PtAccelInfo, PtCamera, PtConfig, PtGraphicsSettings, PtRect, PtSceneData, PtStats)

_PKG = os.path.dirname(os.path.abspath(__file__))

STATUS = {0: "PT_OK", 1: "PT_ERR_INVALID_ARG", 2: "PT_ERR_NO_DEVICE", 3: "PT_ERR_HIP", 4: "PT_ERR_STATE", 5: "PT_ERR_UNSUPPORTED", 6: "PT_ERR_OOM"}

# every symbol include/pt_api.h declares
API_SYMBOLS = [
    "pt_create", "pt_destroy", "pt_set_scene", "pt_build_accel", "pt_update_spheres", "pt_refit_accel", "pt_set_camera", "pt_set_constants", "pt_render",
    "pt_set_partition", "pt_tiles_count", "pt_render_tiles", "pt_unpack_tiles", "pt_set_partition_ex", "pt_tiles_count_ex", "pt_unpack_tiles_ex", "pt_tonemap", "pt_accumulate", "pt_set_textures", "pt_update_rotations", "pt_pack_rgb", "pt_unpack_tiles_rgb", "pt_trace_rays", "pt_trace_rays_stats", "pt_accel_download",
    "pt_accel_download_order", "pt_lbvh_build_host", "pt_sah_build_host", "pt_set_profiling", "pt_get_profile", "pt_get_totals", "pt_get_queue_sizes", "pt_synchronize", "pt_last_error", "pt_version",
    "pt_comm_unique_id", "pt_comm_init", "pt_comm_destroy", "pt_gather", "pt_device_alloc", "pt_device_free", "pt_download",
]


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS.get(status, status)}: {message}")
        self.status = status


def hip_library_path():
    # PT_HIP_LIB: another build of the same library (A/B runs of kernel variants on one box: tools/experiments)
    return os.environ.get("PT_HIP_LIB") or os.path.join(_PKG, "libpt_hip.so")


class HipLib:
    def __init__(self, path=None):
        path = path or hip_library_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: the HIP extension is required (no CPU fallback)")
        self.lib = lib = C.CDLL(path)
        vp, u32 = C.c_void_p, C.c_uint32
        lib.pt_create.restype = C.c_int
        lib.pt_create.argtypes = [C.POINTER(PtConfig), C.POINTER(vp)]
        lib.pt_destroy.restype = None
        lib.pt_destroy.argtypes = [vp]
        lib.pt_set_scene.restype = C.c_int
        lib.pt_set_scene.argtypes = [vp, vp, vp, u32, C.POINTER(PtSceneData)]
        lib.pt_build_accel.restype = C.c_int
        lib.pt_build_accel.argtypes = [vp, C.POINTER(PtAccelInfo)]
        lib.pt_update_spheres.restype = C.c_int
        lib.pt_update_spheres.argtypes = [vp, vp, u32]
        lib.pt_refit_accel.restype = C.c_int
        lib.pt_refit_accel.argtypes = [vp]
        lib.pt_set_camera.restype = C.c_int
        lib.pt_set_camera.argtypes = [vp, C.POINTER(PtCamera)]
        lib.pt_set_constants.restype = C.c_int
        lib.pt_set_constants.argtypes = [vp, C.POINTER(PtGraphicsSettings)]
        lib.pt_render.restype = C.c_int
        lib.pt_render.argtypes = [vp, C.POINTER(PtRect), vp, C.c_int, C.POINTER(PtStats)]
        lib.pt_set_partition.restype = C.c_int
        lib.pt_set_partition.argtypes = [vp, u32, u32]
        lib.pt_tiles_count.restype = u32
        lib.pt_tiles_count.argtypes = [vp, u32]
        lib.pt_render_tiles.restype = C.c_int
        lib.pt_render_tiles.argtypes = [vp, vp, C.POINTER(PtStats)]
        lib.pt_unpack_tiles.restype = C.c_int
        lib.pt_unpack_tiles.argtypes = [vp, vp, u32, vp]
        lib.pt_set_partition_ex.restype = C.c_int
        lib.pt_set_partition_ex.argtypes = [vp, u32, u32, u32]
        lib.pt_tiles_count_ex.restype = u32
        lib.pt_tiles_count_ex.argtypes = [vp, u32, u32, u32]
        lib.pt_unpack_tiles_ex.restype = C.c_int
        lib.pt_unpack_tiles_ex.argtypes = [vp, vp, C.c_uint64, u32, u32, u32, u32, vp]
        lib.pt_pack_rgb.restype = C.c_int
        lib.pt_pack_rgb.argtypes = [vp, vp, C.c_uint64, vp]
        lib.pt_unpack_tiles_rgb.restype = C.c_int
        lib.pt_unpack_tiles_rgb.argtypes = [vp, vp, C.c_uint64, u32, u32, u32, u32, vp]
        lib.pt_set_textures.restype = C.c_int
        lib.pt_set_textures.argtypes = [vp, vp, u32, vp, vp]
        lib.pt_update_rotations.restype = C.c_int
        lib.pt_update_rotations.argtypes = [vp, vp, u32]
        lib.pt_tonemap.restype = C.c_int
        lib.pt_tonemap.argtypes = [vp, vp, u32, vp, vp]
        lib.pt_accumulate.restype = C.c_int
        lib.pt_accumulate.argtypes = [vp, vp, vp, u32, u32]
        lib.pt_trace_rays.restype = C.c_int
        lib.pt_trace_rays.argtypes = [vp, vp, vp, u32, C.c_float, C.c_int, vp, vp]
        lib.pt_trace_rays_stats.restype = C.c_int
        lib.pt_trace_rays_stats.argtypes = [vp, vp, vp, u32, C.c_float, vp, vp, vp]
        lib.pt_accel_download.restype = C.c_int
        lib.pt_accel_download.argtypes = [vp, vp, u32]
        lib.pt_accel_download_order.restype = C.c_int
        lib.pt_accel_download_order.argtypes = [vp, vp, u32]
        lib.pt_lbvh_build_host.restype = C.c_int
        lib.pt_lbvh_build_host.argtypes = [vp, u32, vp, vp, C.POINTER(u32)]
        lib.pt_sah_build_host.restype = C.c_int
        lib.pt_sah_build_host.argtypes = [vp, u32, vp, vp, C.POINTER(u32)]
        lib.pt_set_profiling.restype = C.c_int
        lib.pt_set_profiling.argtypes = [vp, C.c_int]
        lib.pt_get_profile.restype = C.c_int
        lib.pt_get_profile.argtypes = [vp, C.POINTER(PtStats), C.c_int]
        lib.pt_get_totals.restype = C.c_int
        lib.pt_get_totals.argtypes = [vp, C.POINTER(PtStats), C.c_int]
        lib.pt_get_queue_sizes.restype = C.c_int
        lib.pt_get_queue_sizes.argtypes = [vp, vp, u32, C.POINTER(u32)]
        lib.pt_synchronize.restype = C.c_int
        lib.pt_synchronize.argtypes = [vp]
        lib.pt_comm_unique_id.restype = C.c_int
        lib.pt_comm_unique_id.argtypes = [vp]
        lib.pt_comm_init.restype = C.c_int
        lib.pt_comm_init.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
        lib.pt_comm_destroy.restype = C.c_int
        lib.pt_comm_destroy.argtypes = [vp]
        lib.pt_gather.restype = C.c_int
        lib.pt_gather.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32]
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [vp]
        lib.pt_version.restype = C.c_char_p
        lib.pt_version.argtypes = []


    def lbvh_build_host(self, spheres, sah=False):
        """Host LBVH builder, or with sah=True the host SAH builder of small scenes (no GPU needed)
        -> (nodes[BVH_NODE_DTYPE], sorted_id, depth)."""
        spheres = np.ascontiguousarray(spheres)
        n = len(spheres)
        nodes = np.zeros(max(n - 1, 0), dtype=BVH_NODE_DTYPE)
        order = np.zeros(n, dtype=np.uint32)
        depth = C.c_uint32(0)
        fn = self.lib.pt_sah_build_host if sah else self.lib.pt_lbvh_build_host
        st = fn(spheres.ctypes.data, n, nodes.ctypes.data if n > 1 else None, order.ctypes.data, C.byref(depth))
        if st != 0:
            raise PtError(st, "pt_lbvh_build_host")
        return nodes, order, depth.value


_hip = None


def load_hip():
    """Load libpt_hip.so once.  If torch is already imported, let it initialise its (bundled) ROCm runtime first:
    loading /opt/rocm's runtime before torch's makes torch report "No HIP GPUs are available" later."""
    global _hip
    if _hip is None:
        import sys

        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:
                pass
        _hip = HipLib()
    return _hip


class Renderer:
    """One PtContext.  Methods mirror the C-ABI one to one; errors raise PtError with pt_last_error()."""

    def __init__(self, device=0, flags=0, stream=0, tile_size=0, lib=None, frames_in_flight=0):
        self._lib = (lib or load_hip()).lib
        cfg = PtConfig(device=device, tile_size=tile_size, stream=stream, flags=flags, frames_in_flight=frames_in_flight)
        ctx = C.c_void_p()
        st = self._lib.pt_create(C.byref(cfg), C.byref(ctx))
        if st != 0:
            raise PtError(st, "pt_create failed (a HIP device is required; there is no CPU fallback)")
        self._ctx = ctx
        self.accel = None
        self._gs = None

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise PtError(st, self._lib.pt_last_error(self._ctx).decode())

    def set_scene(self, spheres, materials, scene_data, build=True):
        spheres = np.ascontiguousarray(spheres)
        materials = np.ascontiguousarray(materials)
        assert spheres.dtype.itemsize == 16 and materials.dtype.itemsize == 64 and len(spheres) == len(materials)
        self._check(self._lib.pt_set_scene(self._ctx, spheres.ctypes.data, materials.ctypes.data, len(spheres), C.byref(scene_data)))
        return self.build_accel() if build else None

    def build_accel(self):
        info = PtAccelInfo()
        self._check(self._lib.pt_build_accel(self._ctx, C.byref(info)))
        self.accel = info
        return info

    def update_spheres(self, spheres, refit=True):
        """New centres / radii for the same objects, applied to the next frame (asynchronous); refit the LBVH boxes."""
        spheres = np.ascontiguousarray(spheres)
        assert spheres.dtype.itemsize == 16
        self._check(self._lib.pt_update_spheres(self._ctx, spheres.ctypes.data, len(spheres)))
        if refit:
            self._check(self._lib.pt_refit_accel(self._ctx))

    def set_camera(self, camera):
        self._check(self._lib.pt_set_camera(self._ctx, C.byref(camera)))

    def set_constants(self, gs):
        self._check(self._lib.pt_set_constants(self._ctx, C.byref(gs)))
        self._gs = gs

    def set_partition(self, rank, world):
        self._check(self._lib.pt_set_partition(self._ctx, rank, world))

    def tiles_count(self, rank):
        return int(self._lib.pt_tiles_count(self._ctx, rank))

    def set_partition_ex(self, first, run, stride):
        """own the tiles t with first <= t % stride < first + run (weighted partition; see include/pt_api.h)"""
        self._check(self._lib.pt_set_partition_ex(self._ctx, first, run, stride))

    def tiles_count_ex(self, first, run, stride):
        return int(self._lib.pt_tiles_count_ex(self._ctx, first, run, stride))

    def set_profiling(self, enabled):
        self._check(self._lib.pt_set_profiling(self._ctx, 1 if enabled else 0))

    def profile(self, reset=False):
        """Summed per-launch event times over every render call since profiling was switched on (synchronises)."""
        stats = PtStats()
        self._check(self._lib.pt_get_profile(self._ctx, C.byref(stats), 1 if reset else 0))
        return stats

    def totals(self, reset=False):
        """Device-accumulated totals over all render calls since the last reset (synchronises)."""
        stats = PtStats()
        self._check(self._lib.pt_get_totals(self._ctx, C.byref(stats), 1 if reset else 0))
        return stats

    def queue_sizes(self):
        """Ray-queue sizes of the last spp == 1 frame: [slots, rays at bounce 1, rays at bounce 2, ...]."""
        buf = np.zeros(256, dtype=np.uint32)
        n = C.c_uint32(0)
        self._check(self._lib.pt_get_queue_sizes(self._ctx, buf.ctypes.data, len(buf), C.byref(n)))
        return [int(x) for x in buf[: n.value]]

    def synchronize(self):
        self._check(self._lib.pt_synchronize(self._ctx))

    # ---- multi-GPU exchange (pt_comm_* / pt_gather): RCCL behind the C-ABI
    def comm_unique_id(self):
        """128-byte communicator id (rank 0 makes it and ships it to the other ranks)"""
        buf = (C.c_ubyte * 128)()
        st = self._lib.pt_comm_unique_id(buf)
        if st != 0:
            raise PtError(st, "pt_comm_unique_id: RCCL could not be loaded" if st == 5 else "pt_comm_unique_id")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._check(self._lib.pt_comm_init(self._ctx, buf, rank, world))

    def comm_destroy(self):
        self._check(self._lib.pt_comm_destroy(self._ctx))

    def gather(self, send_ptr, recv_ptr, nbytes, root=0):
        """every rank but `root` sends nbytes from send_ptr; the root receives world - 1 parts into recv_ptr (device pointers)"""
        self._check(self._lib.pt_gather(self._ctx, C.c_void_p(send_ptr or 0), C.c_void_p(recv_ptr or 0), int(nbytes), int(root)))

    def render(self, rect=None, want_stats=True):
        """Render to a host numpy array (h, w, 4) float32.  rect = (x, y, w, h) or None for the full frame."""
        if rect is None:
            rect = (0, 0, self._gs.RenderSize[0], self._gs.RenderSize[1])
        r = PtRect(*rect)
        out = np.empty((r.h, r.w, 4), dtype=np.float32)
        stats = PtStats()
        self._check(self._lib.pt_render(self._ctx, C.byref(r), out.ctypes.data, 0, C.byref(stats) if want_stats else None))
        return out, stats

    def render_device(self, out_ptr, rect=None, want_stats=False):
        """Render into device memory (e.g. a torch CUDA tensor's data_ptr()); asynchronous unless want_stats."""
        r = PtRect(*rect) if rect is not None else None
        stats = PtStats()
        self._check(self._lib.pt_render(self._ctx, C.byref(r) if r is not None else None, C.c_void_p(out_ptr), 1, C.byref(stats) if want_stats else None))
        return stats

    def render_tiles(self, out_ptr, want_stats=False):
        stats = PtStats()
        self._check(self._lib.pt_render_tiles(self._ctx, C.c_void_p(out_ptr), C.byref(stats) if want_stats else None))
        return stats

    def unpack_tiles(self, gathered_ptr, max_tiles_per_rank, frame_ptr):
        self._check(self._lib.pt_unpack_tiles(self._ctx, C.c_void_p(gathered_ptr), max_tiles_per_rank, C.c_void_p(frame_ptr)))

    def unpack_tiles_ex(self, packed_ptr, part_stride_px, n_parts, first0, run, stride, frame_ptr):
        self._check(self._lib.pt_unpack_tiles_ex(self._ctx, C.c_void_p(packed_ptr), part_stride_px, n_parts, first0, run, stride, C.c_void_p(frame_ptr)))

    def set_textures(self, texture_set):
        """textures.TextureSet -> pt_set_textures (None removes all textures); call after set_scene"""
        if texture_set is None:
            self._check(self._lib.pt_set_textures(self._ctx, None, 0, None, None))
            return
        tex, n_tex, obj, rot = texture_set.as_ctypes()
        self._check(self._lib.pt_set_textures(self._ctx, C.cast(tex, C.c_void_p), n_tex, C.cast(obj, C.c_void_p), rot.ctypes.data))

    def update_rotations(self, rotations_xyzw):
        q = np.ascontiguousarray(rotations_xyzw, dtype=np.float32).reshape(-1, 4)
        self._check(self._lib.pt_update_rotations(self._ctx, q.ctypes.data, len(q)))

    def tonemap(self, hdr_ptr, n_pixels, params, out_ptr):
        """display transform (row N3): device float4[n] -> device packed uint32[n], asynchronous on the context's stream"""
        self._check(self._lib.pt_tonemap(self._ctx, C.c_void_p(hdr_ptr), n_pixels, C.addressof(params), C.c_void_p(out_ptr)))

    def accumulate(self, accum_ptr, radiance_ptr, n_pixels, frames_accumulated):
        """running mean of successive frames (device pointers), asynchronous on the context's stream"""
        self._check(self._lib.pt_accumulate(self._ctx, C.c_void_p(accum_ptr), C.c_void_p(radiance_ptr), n_pixels, frames_accumulated))

    def pack_rgb(self, src_ptr, n_pixels, dst_ptr):
        """device float4[n] -> device 3 floats per pixel (the 12-byte exchange format)"""
        self._check(self._lib.pt_pack_rgb(self._ctx, C.c_void_p(src_ptr), n_pixels, C.c_void_p(dst_ptr)))

    def unpack_tiles_rgb(self, packed_ptr, part_stride_px, n_parts, first0, run, stride, frame_ptr):
        self._check(self._lib.pt_unpack_tiles_rgb(self._ctx, C.c_void_p(packed_ptr), part_stride_px, n_parts, first0, run, stride, C.c_void_p(frame_ptr)))

    def trace_rays(self, origins, directions, tmin=0.0, use_bvh=True):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, dtype=np.float32)
        ids = np.empty(n, dtype=np.uint32)
        self._check(self._lib.pt_trace_rays(self._ctx, o.ctypes.data, d.ctypes.data, n, tmin, 1 if use_bvh else 0, t.ctypes.data, ids.ctypes.data))
        return t, ids

    def trace_rays_stats(self, origins, directions, tmin=0.0):
        """-> (t, ids, visits) with visits[:, 0] = internal nodes visited, visits[:, 1] = spheres tested per ray"""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        t = np.empty(n, dtype=np.float32)
        ids = np.empty(n, dtype=np.uint32)
        visits = np.zeros((n, 2), dtype=np.uint32)
        self._check(self._lib.pt_trace_rays_stats(self._ctx, o.ctypes.data, d.ctypes.data, n, tmin, t.ctypes.data, ids.ctypes.data, visits.ctypes.data))
        return t, ids, visits

    def download_accel(self):
        n = self.accel.node_count
        nodes = np.zeros(n, dtype=BVH_NODE_DTYPE)
        self._check(self._lib.pt_accel_download(self._ctx, nodes.ctypes.data, n))
        order = np.zeros(self.accel.leaf_count, dtype=np.uint32)
        self._check(self._lib.pt_accel_download_order(self._ctx, order.ctypes.data, len(order)))
        return nodes, order
