/* pt_types.h -- plain-C data layouts of the path-tracing hot path's API surface.
 *
 * These are byte-for-byte the structs the reference host uploads to the GPU for
 * the bounce loop (citations are into /root/reference):
 *
 *   PtMaterial         == Material            Source/Material.ixx:12-20  (== Shaders/Material.hlsli:8-17), 64 B
 *   PtCamera           == Camera              Source/Camera.ixx:16-36    (== Shaders/Camera.hlsli:5-25), 608 B payload
 *   PtSceneData        == SceneData           Source/CommonShaderData.ixx:15-20 (== Shaders/Common.hlsli:7-13), 80 B payload
 *   PtGraphicsSettings == _GraphicsSettings   Source/Raytracing.ixx:151-166 (== Shaders/Raytracing.hlsl:21-39), 80 B
 *
 * PtSphere replaces the reference's per-instance ObjectToWorld of the unit
 * geosphere mesh (Source/Scene.ixx:188-203: scale = 2*radius, z flipped): the
 * build intersects analytic spheres, so an instance is (centre, radius).
 *
 * No torch / HIP types appear here; this header is shared by the C-ABI
 * (include/pt_api.h), the C++ host mirror and the CPU oracle (oracle/).
 */
#ifndef PT_TYPES_H
#define PT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtSphere {
    float cx, cy, cz; /* world-space centre (render space: PhysX z negated, Scene.ixx:197-199) */
    float r;          /* radius > 0 */
} PtSphere;

/* Source/Material.ixx:10 */
enum { PT_ALPHA_OPAQUE = 0, PT_ALPHA_MASK = 1, PT_ALPHA_BLEND = 2 };

/* Source/Material.ixx:12-20; defaults are BaseColor (0,0,0,1), EmissiveStrength 1,
 * EmissiveColor 0, Metallic 0, Roughness 0.5, IOR 1.5, Transmission 0, Opaque, AlphaCutoff 0.5 */
typedef struct PtMaterial {
    float BaseColor[4];      /*  0 */
    float EmissiveStrength;  /* 16 */
    float EmissiveColor[3];  /* 20 */
    float Metallic;          /* 32 */
    float Roughness;         /* 36 */
    float IOR;               /* 40 */
    float Transmission;      /* 44 */
    uint32_t AlphaMode;      /* 48: not Opaque = non-opaque geometry (Source/Scene.ixx:242-243): every crossing of a ray with the sphere */
    float AlphaCutoff;       /* 52: is a candidate that counts only if BaseColor.a (times the base-colour map's alpha) >= AlphaCutoff --
                                    Mask and Blend alike (Shaders/RaytracingHelpers.hlsli:19-43, ShadingHelpers.hlsli:105-115) */
    uint32_t _pad[2];        /* 56 */
} PtMaterial;

/* Source/Camera.ixx:16-36. The bounce loop reads only Position, Right/Up/Forward
 * (un-normalised, lens scaled), NearDepth, FarDepth and Jitter
 * (Shaders/Raytracing.hlsl:114,126,138; Shaders/Camera.hlsli:27-41). */
typedef struct PtCamera {
    uint32_t IsNormalizedDepthReversed; /*   0 */
    float PreviousPosition[3];          /*   4 */
    float Position[3];                  /*  16 */
    float _pad0;                        /*  28 */
    float RightDirection[3];            /*  32 */
    float _pad1;                        /*  44 */
    float UpDirection[3];               /*  48 */
    float _pad2;                        /*  60 */
    float ForwardDirection[3];          /*  64 */
    float ApertureRadius;               /*  76 */
    float NearDepth;                    /*  80 */
    float FarDepth;                     /*  84 */
    float Jitter[2];                    /*  88 */
    float Matrices[8][16];              /*  96: PreviousWorldToView, PreviousViewToProjection,
                                                PreviousWorldToProjection, PreviousProjectionToView,
                                                PreviousViewToWorld, WorldToProjection, ProjectionToView,
                                                ViewToWorld -- unused by the bounce loop */
} PtCamera;

/* Source/CommonShaderData.ixx:15-20. EnvironmentLightColor.a < 0 selects the
 * procedural sky (Shaders/ShadingHelpers.hlsli:25-29).  EnvironmentLightTextureDescriptor
 * != ~0u selects an environment map (ShadingHelpers.hlsli:13-24): it indexes the
 * PtTexture table given to pt_set_textures (this path's descriptor heap), the lookup
 * direction is rotated by the upper 3x3 of EnvironmentLightTransform.  With
 * IsEnvironmentLightTextureCubeMap the descriptor names the first of six consecutive
 * square faces of one size in D3D order (+X, -X, +Y, -Y, +Z, -Z); else a lat-long map. */
typedef struct PtSceneData {
    uint32_t IsStatic;                          /*  0 */
    uint32_t IsEnvironmentLightTextureCubeMap;  /*  4 */
    uint32_t EnvironmentLightTextureDescriptor; /*  8 */
    uint32_t _pad;                              /* 12 */
    float EnvironmentLightColor[4];             /* 16 */
    float EnvironmentLightTransform[12];        /* 32: row-major float3x4 */
} PtSceneData;

/* GPU-side layout of GraphicsSettings, Source/Raytracing.ixx:151-166: HLSL bools are 4 bytes. */
typedef struct PtGraphicsSettings {
    uint32_t RenderSize[2];                      /*  0 */
    uint32_t FrameIndex;                         /*  8 */
    uint32_t Bounces;                            /* 12 */
    uint32_t SamplesPerPixel;                    /* 16 */
    float ThroughputThreshold;                   /* 20: reference default 1e-3 (Raytracing.ixx:33) */
    uint32_t IsRussianRouletteEnabled;           /* 24 */
    uint32_t IsShaderExecutionReorderingEnabled; /* 28: ignored (NV SER has no meaning here) */
    uint32_t IsDIEnabled;                        /* 32: 1 = sphere-light direct illumination of the primary surface (row N4, a stand-in for ReSTIR-DI) */
    uint32_t Denoiser;                           /* 36: must be 0 == Denoiser::None */
    uint32_t _pad0[2];                           /* 40 */
    uint32_t SHARC_Capacity;                     /* 48: SHARC block ignored (dropped) */
    float SHARC_SceneScale;                      /* 52 */
    float SHARC_RoughnessThreshold;              /* 56 */
    uint32_t SHARC_IsAntiFireflyEnabled;         /* 60 */
    uint32_t SHARC_IsHashGridVisualizationEnabled; /* 64 */
    uint32_t _pad1[3];                           /* 68 */
} PtGraphicsSettings;

/* Row N1 -- textured spheres.  TextureMapInfo mirrors Shaders/Material.hlsli:39-43 / Source/Material.ixx:35-38 (16 bytes);
 * Descriptor indexes the PtTexture array given to pt_set_textures instead of a D3D12 descriptor heap. */
typedef struct PtTextureMapInfo {
    uint32_t Descriptor;              /* ~0u = no texture */
    uint32_t TextureCoordinateIndex;  /* spheres have one UV set: must be 0 */
    uint32_t _pad[2];
} PtTextureMapInfo;

enum { PT_TEXTURE_MAP_BASE_COLOR = 0, PT_TEXTURE_MAP_EMISSIVE_COLOR = 1, PT_TEXTURE_MAP_METALLIC = 2, PT_TEXTURE_MAP_ROUGHNESS = 3,
       PT_TEXTURE_MAP_METALLIC_ROUGHNESS = 4, PT_TEXTURE_MAP_TRANSMISSION = 5, PT_TEXTURE_MAP_NORMAL = 6, PT_TEXTURE_MAP_COUNT = 7 };

/* TextureMapInfoArray of one object (Shaders/Common.hlsli:27, ObjectData::TextureMapInfoArray) */
typedef struct PtObjectTextures {
    PtTextureMapInfo Maps[PT_TEXTURE_MAP_COUNT];
} PtObjectTextures;

/* A decoded image as the reference's TextureHelpers hands it to D3D12 (Source/TextureHelpers.ixx:34-60): 8-bit RGBA texels,
 * either linear (DXGI_FORMAT_R8G8B8A8_UNORM) or sRGB-encoded colour (.._UNORM_SRGB, `forceSRGB`); alpha is always linear. */
typedef struct PtTexture {
    const void *Pixels;    /* host pointer, Width * Height texels of 4 bytes (RGBA8) or 16 bytes (RGBA32_FLOAT), row-major, tightly packed */
    uint32_t Width, Height;
    uint32_t Format;       /* PT_TEXTURE_RGBA8_UNORM | PT_TEXTURE_RGBA8_UNORM_SRGB | PT_TEXTURE_RGBA32_FLOAT */
    uint32_t _pad;
} PtTexture;

/* RGBA32_FLOAT: linear HDR texels, what the reference's EXR/HDR environment maps decode to (DXGI_FORMAT_R32G32B32A32_FLOAT) */
enum { PT_TEXTURE_RGBA8_UNORM = 0, PT_TEXTURE_RGBA8_UNORM_SRGB = 1, PT_TEXTURE_RGBA32_FLOAT = 2 };

/* Display transform parameters (row N3): what App::Impl::ToneMap hands to DirectXTK's ToneMapPostProcess
 * (Source/App.cpp:1731-1757; operator / transfer-function pairs created at Source/App.cpp:760-769). */
typedef struct PtToneMapParams {
    uint32_t Operator;         /* ToneMapPostProcess::Operator: 0 None, 1 Saturate, 2 Reinhard, 3 ACESFilmic */
    uint32_t TransferFunction; /* ToneMapPostProcess::TransferFunction: 0 Linear, 1 SRGB, 2 ST2084 */
    float LinearExposure;      /* SetExposure(stops) -> 2^stops (SDR paths) */
    float PaperWhiteNits;      /* SetST2084Parameter (HDR10 path), default 200 (Source/MyAppData.h:316) */
    uint32_t ColorRotation;    /* SetColorRotation: 0 HDTV_to_UHDTV, 1 DCI_P3_D65_to_UHDTV, 2 HDTV_to_DCI_P3_D65 */
    uint32_t _pad[3];
} PtToneMapParams;

/* Pixel rectangle in render-target coordinates. */
typedef struct PtRect {
    uint32_t x, y, w, h;
} PtRect;

#ifdef __cplusplus
} /* extern "C" */
#endif

#ifdef __cplusplus
static_assert(sizeof(PtSphere) == 16, "PtSphere layout");
static_assert(sizeof(PtMaterial) == 64, "PtMaterial layout");
static_assert(sizeof(PtCamera) == 608, "PtCamera layout");
static_assert(sizeof(PtSceneData) == 80, "PtSceneData layout");
static_assert(sizeof(PtGraphicsSettings) == 80, "PtGraphicsSettings layout");
static_assert(sizeof(PtToneMapParams) == 32, "PtToneMapParams layout");
static_assert(sizeof(PtTextureMapInfo) == 16 && sizeof(PtObjectTextures) == 112, "TextureMapInfoArray layout");
#else
_Static_assert(sizeof(PtSphere) == 16, "PtSphere layout");
_Static_assert(sizeof(PtMaterial) == 64, "PtMaterial layout");
_Static_assert(sizeof(PtCamera) == 608, "PtCamera layout");
_Static_assert(sizeof(PtSceneData) == 80, "PtSceneData layout");
_Static_assert(sizeof(PtGraphicsSettings) == 80, "PtGraphicsSettings layout");
#endif

#endif /* PT_TYPES_H */
