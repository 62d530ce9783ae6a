// pt_texture.h -- textured spheres: EvaluateMaterial's texture branches and normal mapping
// (Shaders/ShadingHelpers.hlsli:53-103, 161-235), as device functions that also compile on the host for the leaf parity
// tests.  What the reference gets from its triangle meshes is restated analytically for spheres:
//   * texture coordinates: DirectXTK GeometricPrimitive::CreateGeoSphere (un-vendored; Source/MyScene.ixx:56-88 keeps its
//     UVs): longitude = atan2(n.x, -n.z), latitude = acos(n.y), uv = (1 - (longitude / 2pi + 0.5), latitude / pi), n = the
//     object-space unit normal (recollection of Geometry.cpp; build-frozen, spec S6)
//   * tangent: DirectXMesh ComputeTangentFrame -> direction of increasing u on the sphere: normalize(n.z, 0, -n.x), zero at
//     the poles (PerturbNormal is skipped there, `any(T != 0)` of ShadingHelpers.hlsli:222)
//   * object space: world normal rotated back by the object's rotation quaternion (InstanceData::ObjectToWorld)
//   * sampler: `SampleLevel(g_anisotropicSampler, uv, 0)` = level-0 bilinear, wrap addressing, fp32 weights (spec S7)
//   * atan2 / acos: `atan2_spec`, a fixed polynomial (spec S8) -- the same role as sincos_2pi / pow_spec
// Texels live on the device as linear float4 (an 8-bit source is converted once at upload, sRGB through from_srgb).
#pragma once

#include "pt_bsdf.h"

#if !defined(__HIPCC__)
struct float4 { float x, y, z, w; };  // host build of the leaf tests: HIP's vector type is not available
#endif

namespace pt {

struct TexView {
    const float4* texels;  // row-major, linear RGBA
    uint32_t w, h;
};

// TextureMapType (Shaders/Material.hlsli:24-37)
enum : uint32_t { kMapBaseColor = 0, kMapEmissiveColor = 1, kMapMetallic = 2, kMapRoughness = 3, kMapMetallicRoughness = 4,
                  kMapTransmission = 5, kMapNormal = 6, kMapCount = 7 };
constexpr uint32_t kNoTexture = 0xFFFFFFFFu;

// atan2(y, x) in (-pi, pi]; atan2(0, 0) = 0.  |error| < 2e-5 rad (1e-3 texel on a 2048-wide map).
PT_HD float atan2_spec(float y, float x)
{
    const float ax = pt_abs(x), ay = pt_abs(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;
    const float a = mn / mx;
    const float s = a * a;
    float r = a * pt_fma(s, pt_fma(s, pt_fma(s, pt_fma(s, pt_fma(s, -0.01172120f, 0.05265332f), -0.11643287f), 0.19354346f), -0.33262347f), 0.99997726f);
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return y < 0.0f ? -r : r;
}

struct f2 { float x, y; };

PT_HD f2 sphere_uv(f3 n)
{
    const float lon = atan2_spec(n.x, -n.z);
    const float lat = atan2_spec(pt_sqrt(pt_max(pt_fma(-n.y, n.y, 1.0f), 0.0f)), n.y);  // acos(n.y)
    f2 uv;
    uv.x = 1.0f - pt_fma(lon, 0.15915494309189533577f, 0.5f);
    uv.y = lat * 0.31830988618379067154f;
    return uv;
}

PT_HD f3 sphere_tangent(f3 n)
{
    const float l2 = pt_fma(n.z, n.z, n.x * n.x);
    if (!(l2 > 0.0f)) return make_f3(0.0f, 0.0f, 0.0f);
    const float inv = pt_rcp(pt_sqrt(l2));
    return make_f3(n.z * inv, 0.0f, -n.x * inv);
}

PT_HD f3 cross(f3 a, f3 b) { return make_f3(pt_fma(a.y, b.z, -(a.z * b.y)), pt_fma(a.z, b.x, -(a.x * b.z)), pt_fma(a.x, b.y, -(a.y * b.x))); }

// v rotated by the unit quaternion q = (x, y, z, w): v + w t + q.xyz x t, t = 2 q.xyz x v
PT_HD f3 quat_rotate(float qx, float qy, float qz, float qw, f3 v)
{
    const f3 u = make_f3(qx, qy, qz);
    const f3 t = cross(u, v) * 2.0f;
    return mad(qw, t, v) + cross(u, t);
}

// one channel-quad of `SampleLevel(g_anisotropicSampler, uv, 0)`: bilinear, wrap
PT_HD uint32_t wrap_index(int i, uint32_t n) { const int m = i % (int)n; return (uint32_t)(m < 0 ? m + (int)n : m); }
PT_HD float lerp1(float a, float b, float t) { return pt_fma(t, b - a, a); }

PT_HD void sample_bilinear(const TexView& tv, f2 uv, float out[4])
{
    // keep the texel coordinate inside int range whatever uv is (NaN -> texel 0)
    float u = uv.x, v = uv.y;
    if (!(pt_abs(u) < 65536.0f)) u = 0.0f;
    if (!(pt_abs(v) < 65536.0f)) v = 0.0f;
    const float x = pt_fma(u, (float)tv.w, -0.5f), y = pt_fma(v, (float)tv.h, -0.5f);
    const float xf = pt_floor(x), yf = pt_floor(y);
    const float fx = x - xf, fy = y - yf;
    const uint32_t x0 = wrap_index((int)xf, tv.w), x1 = wrap_index((int)xf + 1, tv.w);
    const uint32_t y0 = wrap_index((int)yf, tv.h), y1 = wrap_index((int)yf + 1, tv.h);
    const float4 c00 = tv.texels[(size_t)y0 * tv.w + x0], c10 = tv.texels[(size_t)y0 * tv.w + x1];
    const float4 c01 = tv.texels[(size_t)y1 * tv.w + x0], c11 = tv.texels[(size_t)y1 * tv.w + x1];
    out[0] = lerp1(lerp1(c00.x, c10.x, fx), lerp1(c01.x, c11.x, fx), fy);
    out[1] = lerp1(lerp1(c00.y, c10.y, fx), lerp1(c01.y, c11.y, fx), fy);
    out[2] = lerp1(lerp1(c00.z, c10.z, fx), lerp1(c01.z, c11.z, fx), fy);
    out[3] = lerp1(lerp1(c00.w, c10.w, fx), lerp1(c01.w, c11.w, fx), fy);
}

// GetEnvironmentLightColor's texture branch (ShadingHelpers.hlsli:13-24) for a lat-long map: rotate by the upper 3x3 of
// EnvironmentLightTransform (Geometry::RotateVector = mul(M, v)), normalise, Math::ToLatLongCoordinate (Math.hlsli:29-33:
// u = (1 + atan2(x, z) / pi) / 2, v = acos(y) / pi), level-0 sample.
PT_HD f2 latlong_uv(f3 d)
{
    f2 uv;
    uv.x = pt_fma(atan2_spec(d.x, d.z), 0.15915494309189533577f, 0.5f);
    uv.y = atan2_spec(pt_sqrt(pt_max(pt_fma(-d.y, d.y, 1.0f), 0.0f)), d.y) * 0.31830988618379067154f;
    return uv;
}

PT_HD f3 environment_texture(const TexView& tv, const float* m, f3 d)
{
    const f3 r = normalize(make_f3(dot(make_f3(m[0], m[1], m[2]), d), dot(make_f3(m[3], m[4], m[5]), d), dot(make_f3(m[6], m[7], m[8]), d)));
    float s[4];
    sample_bilinear(tv, latlong_uv(r), s);
    return make_f3(s[0], s[1], s[2]);
}

// ... and for a cube map, `TextureCube::SampleLevel(sampler, d, 0)` (spec S9, build-frozen): the face is the axis of largest
// magnitude (ties: z over y over x), faces in D3D order +X, -X, +Y, -Y, +Z, -Z, face coordinates (sc, tc) / |major| mapped
// to [0, 1]^2 by the D3D table below, level-0 bilinear INSIDE the face with clamp addressing (hardware filters across face
// edges; the difference is confined to the outermost half texel of a face).
struct CubeCoord { uint32_t face; f2 uv; };

PT_HD CubeCoord cube_face_uv(f3 d)
{
    const float ax = pt_abs(d.x), ay = pt_abs(d.y), az = pt_abs(d.z);
    CubeCoord c;
    float sc, tc, ma;
    if (az >= ax && az >= ay) { c.face = d.z < 0.0f ? 5u : 4u; sc = d.z < 0.0f ? -d.x : d.x; tc = -d.y; ma = az; }
    else if (ay >= ax)        { c.face = d.y < 0.0f ? 3u : 2u; sc = d.x; tc = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else                      { c.face = d.x < 0.0f ? 1u : 0u; sc = d.x < 0.0f ? d.z : -d.z; tc = -d.y; ma = ax; }
    c.uv.x = pt_fma(sc / ma, 0.5f, 0.5f);
    c.uv.y = pt_fma(tc / ma, 0.5f, 0.5f);
    return c;
}

PT_HD uint32_t clamp_index(int i, uint32_t n) { return i < 0 ? 0u : ((uint32_t)i >= n ? n - 1u : (uint32_t)i); }

PT_HD void sample_bilinear_clamp(const TexView& tv, f2 uv, float out[4])
{
    float u = uv.x, v = uv.y;
    if (!(pt_abs(u) < 65536.0f)) u = 0.0f;  // NaN direction (0/0): texel 0
    if (!(pt_abs(v) < 65536.0f)) v = 0.0f;
    const float x = pt_fma(u, (float)tv.w, -0.5f), y = pt_fma(v, (float)tv.h, -0.5f);
    const float xf = pt_floor(x), yf = pt_floor(y);
    const float fx = x - xf, fy = y - yf;
    const uint32_t x0 = clamp_index((int)xf, tv.w), x1 = clamp_index((int)xf + 1, tv.w);
    const uint32_t y0 = clamp_index((int)yf, tv.h), y1 = clamp_index((int)yf + 1, tv.h);
    const float4 c00 = tv.texels[(size_t)y0 * tv.w + x0], c10 = tv.texels[(size_t)y0 * tv.w + x1];
    const float4 c01 = tv.texels[(size_t)y1 * tv.w + x0], c11 = tv.texels[(size_t)y1 * tv.w + x1];
    out[0] = lerp1(lerp1(c00.x, c10.x, fx), lerp1(c01.x, c11.x, fx), fy);
    out[1] = lerp1(lerp1(c00.y, c10.y, fx), lerp1(c01.y, c11.y, fx), fy);
    out[2] = lerp1(lerp1(c00.z, c10.z, fx), lerp1(c01.z, c11.z, fx), fy);
    out[3] = lerp1(lerp1(c00.w, c10.w, fx), lerp1(c01.w, c11.w, fx), fy);
}

// faces: six consecutive entries of the texture table
PT_HD f3 environment_cube(const TexView* faces, const float* m, f3 d)
{
    const f3 r = normalize(make_f3(dot(make_f3(m[0], m[1], m[2]), d), dot(make_f3(m[3], m[4], m[5]), d), dot(make_f3(m[6], m[7], m[8]), d)));
    const CubeCoord c = cube_face_uv(r);
    float s[4];
    sample_bilinear_clamp(faces[c.face], c.uv, s);
    return make_f3(s[0], s[1], s[2]);
}

// Geometry::UnpackLocalNormal (MathLib, un-vendored; recollection): xy = s * 255/127 - 1, z = Sqrt01(1 - |xy|^2)
PT_HD f3 unpack_local_normal(float sx, float sy)
{
    const float k = 255.0f / 127.0f;
    const float x = pt_fma(sx, k, -1.0f), y = pt_fma(sy, k, -1.0f);
    return make_f3(x, y, sqrt01(1.0f - pt_fma(y, y, x * x)));
}

// PerturbNormal (ShadingHelpers.hlsli:87-103) with Math::CalculateTBN (Math.hlsli:17-21)
PT_HD f3 perturb_normal(f3 N, f3 T, float sx, float sy)
{
    const f3 nl = unpack_local_normal(sx, sy);
    const f3 Tn = normalize(T - N * dot(N, T));
    const f3 B = cross(N, Tn);
    // RotateVectorInverse(TBN, nl) = nl.x T + nl.y B + nl.z N
    return normalize(mad(nl.x, Tn, mad(nl.y, B, N * nl.z)));
}

// The material after EvaluateMaterial (ShadingHelpers.hlsli:161-235); alpha is not used by the path (opaque spheres)
struct MaterialEval {
    f3 BaseColor, EmissiveColor;
    float EmissiveStrength, Metallic, Roughness, Transmission;
    f3 Ns;  // shading normal after PerturbNormal
};

// maps[kMapCount]: texture index per TextureMapType (kNoTexture = none).  Ns_in: the (front-facing) shading normal,
// T: the front tangent (zero = no normal mapping), uv: the hit's texture coordinates.
PT_HD MaterialEval evaluate_material(const TexView* tex, const uint32_t* maps, f2 uv, f3 base, float emissive_strength, f3 emissive_color,
                                     float metallic, float roughness, float transmission, f3 Ns_in, f3 T)
{
    MaterialEval m;
    m.BaseColor = base; m.EmissiveColor = emissive_color; m.EmissiveStrength = emissive_strength;
    m.Metallic = metallic; m.Roughness = roughness; m.Transmission = transmission; m.Ns = Ns_in;
    // One pass over the map types in TextureMapType order, ONE instance of the sampler (the seven inlined copies of the first
    // version cost registers and instruction cache); the conditions and the order of application are the reference's.
    const f3 emission = emissive_color * emissive_strength;
    const bool has_mr = maps[kMapMetallicRoughness] != kNoTexture;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (uint32_t k = 0; k < kMapCount; k++) {
        const uint32_t t = maps[k];
        if (t == kNoTexture) continue;
        bool need;
        switch (k) {
            case kMapBaseColor: need = base.x > 0.0f || base.y > 0.0f || base.z > 0.0f; break;  // :61-72 (the reference also tests alpha > 0; with rgb == 0 the product is 0 either way)
            case kMapEmissiveColor: need = emission.x > 0.0f || emission.y > 0.0f || emission.z > 0.0f; break;  // :178-184
            case kMapMetallic: need = !has_mr && metallic > 0.0f; break;                                  // :197-213
            case kMapRoughness: need = !has_mr && roughness > 0.0f; break;
            case kMapMetallicRoughness: need = metallic > 0.0f || roughness > 0.0f; break;                  // :186-196
            case kMapTransmission: need = m.Metallic < 1.0f && transmission > 0.0f; break;                  // :215-221, :74-85
            default: need = T.x != 0.0f || T.y != 0.0f || T.z != 0.0f; break;                               // kMapNormal, :222-230
        }
        if (!need) continue;
        float s[4];
        sample_bilinear(tex[t], uv, s);
        switch (k) {
            case kMapBaseColor: m.BaseColor = make_f3(base.x * s[0], base.y * s[1], base.z * s[2]); break;
            case kMapEmissiveColor: m.EmissiveColor = make_f3(emissive_color.x * s[0], emissive_color.y * s[1], emissive_color.z * s[2]); break;
            case kMapMetallic: m.Metallic = metallic * s[0]; break;
            case kMapRoughness: m.Roughness = roughness * s[0]; break;
            case kMapMetallicRoughness: m.Metallic = metallic * s[2]; m.Roughness = roughness * s[1]; break;
            case kMapTransmission: m.Transmission = transmission * s[0]; break;
            default: m.Ns = perturb_normal(Ns_in, T, s[0], s[1]); break;
        }
    }
    return m;
}

}  // namespace pt
