"""Texture sets for textured spheres (row N1): decoded 8-bit RGBA images + one TextureMapInfoArray per sphere + object
rotations, in the layout pt_set_textures (include/pt_api.h) and the oracle take.  Also a few procedural images: the
reference's own assets (Assets/Textures/*.png, *.jpg) cannot travel with this repository, so the demo scene's textured
objects (Alien-Metal, Moon, Earth: Source/MyScene.ixx:161-166, 277-295) get procedural stand-ins; `load_image` decodes
the real files when a user has them (PIL)."""
import ctypes as C

import numpy as np

from .abi_types import (TEXTURE_MAP_COUNT, TEXTURE_RGBA8_UNORM, TEXTURE_RGBA8_UNORM_SRGB, TEXTURE_RGBA32_FLOAT, PtObjectTextures, PtTexture)

NO_TEXTURE = 0xFFFFFFFF


class TextureSet:
    def __init__(self, n_spheres):
        self.n = n_spheres
        self.images = []   # (uint8 array (h, w, 4), format)
        self.maps = np.full((n_spheres, TEXTURE_MAP_COUNT), NO_TEXTURE, dtype=np.uint32)
        self.rotations = np.tile(np.array([0, 0, 0, 1], dtype=np.float32), (n_spheres, 1))
        self._keep = None

    def add_image(self, rgba, srgb=False):
        """rgba: uint8 (h, w, 4) (or (h, w, 3) / (h, w): alpha 255, grey replicated); returns the texture index"""
        a = np.asarray(rgba, dtype=np.uint8)
        if a.ndim == 2:
            a = np.repeat(a[..., None], 3, -1)
        if a.shape[-1] == 3:
            a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], -1)
        self.images.append((np.ascontiguousarray(a), TEXTURE_RGBA8_UNORM_SRGB if srgb else TEXTURE_RGBA8_UNORM))
        return len(self.images) - 1

    def add_hdr_image(self, rgba):
        """rgba: float32 (h, w, 4) or (h, w, 3) linear HDR texels (PT_TEXTURE_RGBA32_FLOAT), e.g. a lat-long environment map
        that SceneData.EnvironmentLightTextureDescriptor then names by the returned index"""
        a = np.asarray(rgba, dtype=np.float32)
        if a.shape[-1] == 3:
            a = np.concatenate([a, np.ones(a.shape[:2] + (1,), np.float32)], -1)
        self.images.append((np.ascontiguousarray(a), TEXTURE_RGBA32_FLOAT))
        return len(self.images) - 1

    def add_cube(self, faces):
        """six square float32 (s, s, 3|4) faces in D3D order (+X, -X, +Y, -Y, +Z, -Z) -> index of the first one: the
        descriptor of a cube-map environment light (IsEnvironmentLightTextureCubeMap = 1)"""
        assert len(faces) == 6
        first = self.add_hdr_image(faces[0])
        for f in faces[1:]:
            self.add_hdr_image(f)
        return first

    def assign(self, sphere, map_type, texture_index):
        self.maps[sphere, map_type] = texture_index

    def set_rotation(self, sphere, quaternion_xyzw):
        q = np.asarray(quaternion_xyzw, dtype=np.float64)
        self.rotations[sphere] = (q / np.linalg.norm(q)).astype(np.float32)

    def as_ctypes(self):
        """(PtTexture array, n_textures, PtObjectTextures array, rotations float32 (n, 4)); the arrays stay alive with self"""
        tex = (PtTexture * max(len(self.images), 1))()
        for i, (img, fmt) in enumerate(self.images):
            tex[i].Pixels = img.ctypes.data
            tex[i].Height, tex[i].Width = img.shape[:2]
            tex[i].Format = fmt
        obj = (PtObjectTextures * max(self.n, 1))()
        for i in range(self.n):
            for k in range(TEXTURE_MAP_COUNT):
                obj[i].Maps[k].Descriptor = int(self.maps[i, k])
        rot = np.ascontiguousarray(self.rotations, dtype=np.float32)
        self._keep = (tex, obj, rot)
        return tex, len(self.images), obj, rot


def quaternion_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(angle / 2), [np.cos(angle / 2)]]).astype(np.float32)


# ---- procedural images (seeded, deterministic) ---------------------------------------------------------------------------

def checker(w, h, cells=16, a=(230, 230, 230), b=(40, 60, 160)):
    y, x = np.mgrid[0:h, 0:w]
    m = ((x * cells // w) + (y * cells * 2 // h // 2 * 1)) % 2 == 0
    return np.where(m[..., None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)


def value_noise(w, h, seed, octaves=5):
    """tileable fractal value noise in [0, 1] (float64)"""
    rng = np.random.default_rng(seed)
    out = np.zeros((h, w))
    amp, total = 1.0, 0.0
    for o in range(octaves):
        n = 4 << o
        g = rng.random((n, n))
        ys, xs = np.linspace(0, n, h, endpoint=False), np.linspace(0, n, w, endpoint=False)
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        fy, fx = fy * fy * (3 - 2 * fy), fx * fx * (3 - 2 * fx)
        g00 = g[y0[:, None] % n, x0[None, :] % n]; g10 = g[y0[:, None] % n, (x0[None, :] + 1) % n]
        g01 = g[(y0[:, None] + 1) % n, x0[None, :] % n]; g11 = g[(y0[:, None] + 1) % n, (x0[None, :] + 1) % n]
        out += amp * ((g00 * (1 - fx) + g10 * fx) * (1 - fy) + (g01 * (1 - fx) + g11 * fx) * fy)
        total += amp
        amp *= 0.5
    return out / total


def planet_albedo(w, h, seed, land=(70, 120, 50), sea=(20, 50, 130), level=0.5):
    n = value_noise(w, h, seed)
    m = (n > level)[..., None]
    shade = (0.6 + 0.8 * n)[..., None]
    return np.clip(np.where(m, np.array(land) * shade, np.array(sea) * shade), 0, 255).astype(np.uint8)


def normal_map_from_height(height, strength=4.0):
    """tangent-space normal map (xy in RG, encoded s = (n + 1) * 127 / 255 as Geometry::UnpackLocalNormal expects)"""
    hgt = np.asarray(height, dtype=np.float64)
    dx = (np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)) * 0.5 * strength * hgt.shape[1] / 64
    dy = (np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)) * 0.5 * strength * hgt.shape[0] / 64
    nrm = np.stack([-dx, -dy, np.ones_like(hgt)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    enc = np.clip(np.rint((nrm + 1.0) * 127.0), 0, 254)
    return np.concatenate([enc[..., :2], np.full(hgt.shape + (1,), 255.0)], -1).astype(np.uint8)


def sky_latlong(w, h, seed=0, sun_dir=(0.4, 0.6, 0.7), sun_radiance=40.0):
    """a procedural HDR lat-long environment map, float32 (h, w, 3): horizon-to-zenith gradient, tileable clouds, a ground
    tint and a small bright sun -- the stand-in for the reference's Assets/Textures/*.exr|hdr environment (not shippable).
    Texel (x, y) covers u = (x + .5) / w, v = (y + .5) / h of Math::ToLatLongCoordinate."""
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    theta, phi = v * np.pi, (2 * u - 1) * np.pi          # v = acos(y) / pi, u = (1 + atan2(x, z) / pi) / 2
    d = np.stack([np.sin(theta) * np.sin(phi), np.cos(theta), np.sin(theta) * np.cos(phi)], -1)
    t = np.clip(d[..., 1], 0, 1)[..., None]
    sky = (1 - t) * np.array([0.9, 0.9, 0.95]) + t * np.array([0.15, 0.35, 0.9])
    clouds = np.clip(value_noise(w, h, seed, octaves=4) - 0.5, 0, 1)[..., None] * 2.0 * np.clip(d[..., 1:2] * 3, 0, 1)
    ground = np.array([0.25, 0.22, 0.2]) * (1 + 0.3 * value_noise(w, h, seed + 1, octaves=3)[..., None])
    img = np.where(d[..., 1:2] >= 0, sky + clouds, ground)
    s = np.asarray(sun_dir, np.float64); s = s / np.linalg.norm(s)
    img = img + sun_radiance * np.clip((d @ s - 0.995) / 0.005, 0, 1)[..., None] ** 2
    return img.astype(np.float32)


def cube_directions(size):
    """unit direction through every texel centre of the six faces, (6, size, size, 3) float64: the inverse of the D3D face
    table (face, u, v) -> (sc, tc) = (2u - 1, 2v - 1)"""
    t = (np.arange(size) + 0.5) / size * 2 - 1
    tc, sc = np.meshgrid(t, t, indexing="ij")  # rows = v (tc), columns = u (sc)
    one = np.ones_like(sc)
    d = np.stack([np.stack([one, -tc, -sc], -1), np.stack([-one, -tc, sc], -1),     # +X: sc = -z, tc = -y; -X: sc = +z
                  np.stack([sc, one, tc], -1), np.stack([sc, -one, -tc], -1),       # +Y: sc = x, tc = z;   -Y: tc = -z
                  np.stack([sc, -tc, one], -1), np.stack([-sc, -tc, -one], -1)])    # +Z: sc = x, tc = -y;  -Z: sc = -x
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def cube_from_function(size, fn):
    """six faces (size, size, 3) float32 of fn(directions (..., 3)) -> (..., 3)"""
    return [fn(d).astype(np.float32) for d in cube_directions(size)]


def load_image(path, srgb_hint=None):
    """decode a PNG / JPG (e.g. the reference's Assets/Textures) -> uint8 (h, w, 4)"""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
