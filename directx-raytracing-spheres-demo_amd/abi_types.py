"""ctypes / numpy mirrors of include/pt_types.h and include/pt_api.h."""
import ctypes as C

import numpy as np


class PtSphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("r", C.c_float)]


class PtMaterial(C.Structure):
    _fields_ = [
        ("BaseColor", C.c_float * 4), ("EmissiveStrength", C.c_float), ("EmissiveColor", C.c_float * 3),
        ("Metallic", C.c_float), ("Roughness", C.c_float), ("IOR", C.c_float), ("Transmission", C.c_float),
        ("AlphaMode", C.c_uint32), ("AlphaCutoff", C.c_float), ("_pad", C.c_uint32 * 2),
    ]


class PtCamera(C.Structure):
    _fields_ = [
        ("IsNormalizedDepthReversed", C.c_uint32), ("PreviousPosition", C.c_float * 3), ("Position", C.c_float * 3),
        ("_pad0", C.c_float), ("RightDirection", C.c_float * 3), ("_pad1", C.c_float), ("UpDirection", C.c_float * 3),
        ("_pad2", C.c_float), ("ForwardDirection", C.c_float * 3), ("ApertureRadius", C.c_float),
        ("NearDepth", C.c_float), ("FarDepth", C.c_float), ("Jitter", C.c_float * 2), ("Matrices", (C.c_float * 16) * 8),
    ]


class PtSceneData(C.Structure):
    _fields_ = [
        ("IsStatic", C.c_uint32), ("IsEnvironmentLightTextureCubeMap", C.c_uint32),
        ("EnvironmentLightTextureDescriptor", C.c_uint32), ("_pad", C.c_uint32),
        ("EnvironmentLightColor", C.c_float * 4), ("EnvironmentLightTransform", C.c_float * 12),
    ]


class PtGraphicsSettings(C.Structure):
    _fields_ = [
        ("RenderSize", C.c_uint32 * 2), ("FrameIndex", C.c_uint32), ("Bounces", C.c_uint32), ("SamplesPerPixel", C.c_uint32),
        ("ThroughputThreshold", C.c_float), ("IsRussianRouletteEnabled", C.c_uint32),
        ("IsShaderExecutionReorderingEnabled", C.c_uint32), ("IsDIEnabled", C.c_uint32), ("Denoiser", C.c_uint32),
        ("_pad0", C.c_uint32 * 2), ("SHARC_Capacity", C.c_uint32), ("SHARC_SceneScale", C.c_float),
        ("SHARC_RoughnessThreshold", C.c_float), ("SHARC_IsAntiFireflyEnabled", C.c_uint32),
        ("SHARC_IsHashGridVisualizationEnabled", C.c_uint32), ("_pad1", C.c_uint32 * 3),
    ]


class PtRect(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]


class PtTextureMapInfo(C.Structure):
    _fields_ = [("Descriptor", C.c_uint32), ("TextureCoordinateIndex", C.c_uint32), ("_pad", C.c_uint32 * 2)]


TEXTURE_MAP_BASE_COLOR, TEXTURE_MAP_EMISSIVE_COLOR, TEXTURE_MAP_METALLIC, TEXTURE_MAP_ROUGHNESS = 0, 1, 2, 3
TEXTURE_MAP_METALLIC_ROUGHNESS, TEXTURE_MAP_TRANSMISSION, TEXTURE_MAP_NORMAL, TEXTURE_MAP_COUNT = 4, 5, 6, 7
TEXTURE_RGBA8_UNORM, TEXTURE_RGBA8_UNORM_SRGB, TEXTURE_RGBA32_FLOAT = 0, 1, 2


class PtObjectTextures(C.Structure):
    _fields_ = [("Maps", PtTextureMapInfo * TEXTURE_MAP_COUNT)]


class PtTexture(C.Structure):
    _fields_ = [("Pixels", C.c_void_p), ("Width", C.c_uint32), ("Height", C.c_uint32), ("Format", C.c_uint32), ("_pad", C.c_uint32)]


class PtToneMapParams(C.Structure):
    _fields_ = [("Operator", C.c_uint32), ("TransferFunction", C.c_uint32), ("LinearExposure", C.c_float), ("PaperWhiteNits", C.c_float),
                ("ColorRotation", C.c_uint32), ("_pad", C.c_uint32 * 3)]


TONE_NONE, TONE_SATURATE, TONE_REINHARD, TONE_ACES_FILMIC = 0, 1, 2, 3          # DirectX::ToneMapPostProcess::Operator
TRANSFER_LINEAR, TRANSFER_SRGB, TRANSFER_ST2084 = 0, 1, 2                       # ::TransferFunction
ROTATE_709_TO_2020, ROTATE_P3D65_TO_2020, ROTATE_709_TO_P3D65 = 0, 1, 2          # ::ColorPrimaryRotation


def tonemap_params(operator=TONE_ACES_FILMIC, transfer=TRANSFER_SRGB, exposure_stops=0.0, paper_white_nits=200.0, rotation=ROTATE_709_TO_2020):
    """the reference's defaults (Source/MyAppData.h:313-330): ACESFilmic + sRGB at exposure 0 for SDR, 200 nits / HDTV_to_UHDTV for HDR10"""
    p = PtToneMapParams()
    p.Operator, p.TransferFunction, p.ColorRotation = operator, transfer, rotation
    p.LinearExposure = 2.0 ** exposure_stops  # SetExposure: linear exposure = 2^stops
    p.PaperWhiteNits = paper_white_nits
    return p


class PtConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("tile_size", C.c_uint32), ("stream", C.c_uint64), ("flags", C.c_uint32), ("frames_in_flight", C.c_uint32)]


class PtAccelInfo(C.Structure):
    _fields_ = [
        ("leaf_count", C.c_uint32), ("node_count", C.c_uint32), ("depth", C.c_uint32), ("lds_resident", C.c_uint32),
        ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("build_ms", C.c_float), ("builder", C.c_uint32),
    ]


class PtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("paths", C.c_uint64), ("pixels", C.c_uint64), ("ms_total", C.c_double),
        ("ms_traverse", C.c_double), ("ms_shade", C.c_double), ("traverse_launches", C.c_uint32),
        ("shade_launches", C.c_uint32), ("bytes_algorithmic", C.c_uint64),
        ("ms_tail", C.c_double), ("tail_launches", C.c_uint32), ("beams_used", C.c_uint32),
        ("rays_first_pass_inline", C.c_uint64), ("node_visits", C.c_uint64), ("sphere_tests", C.c_uint64),
    ]


class PtBvhNode(C.Structure):
    _fields_ = [
        ("lo0", C.c_float * 3), ("hi0", C.c_float * 3), ("lo1", C.c_float * 3), ("hi1", C.c_float * 3),
        ("child0", C.c_int32), ("child1", C.c_int32), ("parent", C.c_int32), ("_pad", C.c_int32),
    ]


assert C.sizeof(PtSphere) == 16 and C.sizeof(PtMaterial) == 64 and C.sizeof(PtCamera) == 608
assert C.sizeof(PtSceneData) == 80 and C.sizeof(PtGraphicsSettings) == 80 and C.sizeof(PtBvhNode) == 64

SPHERE_DTYPE = np.dtype([("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("r", "<f4")])
MATERIAL_DTYPE = np.dtype([
    ("BaseColor", "<f4", (4,)), ("EmissiveStrength", "<f4"), ("EmissiveColor", "<f4", (3,)), ("Metallic", "<f4"),
    ("Roughness", "<f4"), ("IOR", "<f4"), ("Transmission", "<f4"), ("AlphaMode", "<u4"), ("AlphaCutoff", "<f4"), ("_pad", "<u4", (2,)),
])
BVH_NODE_DTYPE = np.dtype([
    ("lo0", "<f4", (3,)), ("hi0", "<f4", (3,)), ("lo1", "<f4", (3,)), ("hi1", "<f4", (3,)),
    ("child0", "<i4"), ("child1", "<i4"), ("parent", "<i4"), ("_pad", "<i4"),
])
assert SPHERE_DTYPE.itemsize == 16 and MATERIAL_DTYPE.itemsize == 64 and BVH_NODE_DTYPE.itemsize == 64

PT_FLAG_NO_LDS_SCENE = 1
PT_FLAG_NO_GRAPH = 2
PT_FLAG_HOST_LBVH = 4
PT_FLAG_SPLIT_KERNELS = 8
PT_FLAG_TWO_FRAMES_IN_FLIGHT = 16
PT_FLAG_DEFAULT_STREAM = 32
PT_FLAG_FAST_BUILD = 64
PT_BUILDER_DEVICE_LBVH, PT_BUILDER_HOST_LBVH, PT_BUILDER_HOST_SAH = 0, 1, 2


def default_material(n=1):
    """Material defaults of Source/Material.ixx:13-18."""
    m = np.zeros(n, dtype=MATERIAL_DTYPE)
    m["BaseColor"] = (0, 0, 0, 1)
    m["EmissiveStrength"] = 1
    m["Roughness"] = 0.5
    m["IOR"] = 1.5
    m["AlphaCutoff"] = 0.5
    return m


def graphics_settings(width, height, frame_index=0, bounces=8, spp=1, rr=True, threshold=1e-3, di=False):
    """GraphicsSettings with the reference defaults that carry over; di = IsDIEnabled (row N4)."""
    gs = PtGraphicsSettings()
    gs.RenderSize[0], gs.RenderSize[1] = width, height
    gs.FrameIndex, gs.Bounces, gs.SamplesPerPixel = frame_index, bounces, spp
    gs.ThroughputThreshold = threshold
    gs.IsRussianRouletteEnabled = 1 if rr else 0
    gs.IsDIEnabled = 1 if di else 0
    return gs
