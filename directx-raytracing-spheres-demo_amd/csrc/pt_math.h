// pt_math.h -- device leaf arithmetic of the bounce loop (gfx950 HIP; also compilable as
// plain host C++ by tests/ to check each function bit-for-bit against the CPU oracle).
//
// Implements DESIGN.md "Frozen arithmetic spec": the un-vendored NVIDIA MathLib functions the
// reference shaders call (call sites Shaders/BxDF.hlsli, SurfaceVectors.hlsli,
// Raytracing.hlsl) plus the build-defined sincos / pow / ray-sphere / spawn-offset.
// Compile with -ffp-contract=off: fused multiply-adds occur only where pt_fma() is written.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define PT_HD inline
#endif

namespace pt {

struct f3 {
    float x, y, z;
};

PT_HD f3 make_f3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_HD f3 operator+(f3 a, f3 b) { return make_f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD f3 operator-(f3 a, f3 b) { return make_f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD f3 operator*(f3 a, f3 b) { return make_f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD f3 operator*(f3 a, float s) { return make_f3(a.x * s, a.y * s, a.z * s); }
PT_HD f3 operator-(f3 a) { return make_f3(-a.x, -a.y, -a.z); }
// s*a + b, one fma per component
PT_HD f3 mad(float s, f3 a, f3 b) { return make_f3(pt_fma(s, a.x, b.x), pt_fma(s, a.y, b.y), pt_fma(s, a.z, b.z)); }
PT_HD float dot(f3 a, f3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }
PT_HD uint32_t as_uint(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float as_float(uint32_t u) { return __builtin_bit_cast(float, u); }
// Square root, reciprocal and 0.5 / x, IEEE correctly rounded (hipcc's default expansions: 16, 11 and 11 instructions, branch-free).
// Shorter forms exist -- v_rcp_f32 + ONE Newton step is the correctly rounded reciprocal of every normal operand whose reciprocal is
// normal, v_sqrt_f32 + the neighbour fix-up without the rescaling is exact from 2^-96 up (all 2^32 operands checked on the GPU,
// tools/experiments/exact_math.hip) -- but the branch to the full expansion that the remaining operands need costs more scalar
// instructions and scheduling freedom than the vector instructions it saves: C2 0.081 -> 0.083 ms, C3 3.12 -> 3.17 (tools/experiments/README.md).
PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_HD float pt_rcp(float x) { return 1.0f / x; }
PT_HD float pt_half_rcp(float x) { return 0.5f / x; }
PT_HD f3 normalize(f3 a) { float inv = pt_rcp(pt_sqrt(dot(a, a))); return a * inv; }
PT_HD float pt_abs(float x) { return __builtin_fabsf(x); }
PT_HD float pt_max(float a, float b) { return a > b ? a : b; }
PT_HD float pt_min(float a, float b) { return a < b ? a : b; }
PT_HD float saturate(float x) { return !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x); }  // NaN -> 0 as HLSL
PT_HD float sqrt01(float x) { return pt_sqrt(saturate(x)); }
PT_HD float sign(float x) { return x >= 0.0f ? 1.0f : -1.0f; }  // Math::Sign, Sign(0) = +1
PT_HD bool is_finite(float x) { return (as_uint(x) & 0x7F800000u) != 0x7F800000u; }
PT_HD float pt_floor(float x) { return __builtin_floorf(x); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kOffsetScale = 1.52587890625e-05f;  // 2^-16: build-defined sphere spawn offset
constexpr float kMinRoughness = 2e-3f;              // BxDF.hlsli:19
constexpr float kInf = __builtin_huge_valf();

// ---------------------------------------------------------------- RNG: Rng::Hash (Raytracing.hlsl:108,330,351)
PT_HD uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
PT_HD uint32_t hash_combine(uint32_t seed, uint32_t v)
{
    return seed ^ (hash32(v) + 0x9E3779B9u + (seed << 6) + (seed >> 2));
}
PT_HD uint32_t rng_init(uint32_t px, uint32_t py, uint32_t frame)
{
    return hash_combine(hash32(frame + 0x035F9F29u), (px << 16) | py);
}
PT_HD uint32_t rng_next(uint32_t& s) { s = hash32(s); return s; }
PT_HD float rng_float(uint32_t& s)  // (0,1]
{
    uint32_t u = rng_next(s);
    return 2.0f - as_float((u >> 9) | 0x3F800000u);
}

// ---------------------------------------------------------------- build-defined sincos(2*pi*u), log2, exp2, pow
PT_HD void sincos_2pi(float u, float& so, float& co)
{
    float x = u * 4.0f;
    float k = pt_floor(x + 0.5f);
    float y = x - k;
    float z = y * 1.57079632679489661923f;
    float z2 = z * z;
    float ps = pt_fma(z2, pt_fma(z2, pt_fma(z2, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    float s = pt_fma(z * z2, ps, z);
    float pc = pt_fma(z2, pt_fma(z2, pt_fma(z2, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    float c = pt_fma(z2 * z2, pc, pt_fma(z2, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    // quadrant rotation without divergent branches
    float a = (q & 1) ? c : s;
    float b = (q & 1) ? s : c;
    so = (q & 2) ? -a : a;
    co = ((q + 1) & 2) ? -b : b;
}

PT_HD float log2_spec(float x)
{
    uint32_t b = as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = pt_fma(s2, pt_fma(s2, pt_fma(s2, pt_fma(s2, 0.111111111f, 0.142857143f), 0.2f), 0.333333333f), 1.0f);
    float ln_m = (2.0f * s) * p;
    return pt_fma(ln_m, 1.44269504088896340736f, (float)e);
}

PT_HD float exp2_spec(float y)
{
    float k = pt_floor(y + 0.5f);
    float f = y - k;
    float t = f * 0.69314718055994530942f;
    float p = pt_fma(t, pt_fma(t, pt_fma(t, pt_fma(t, pt_fma(t, pt_fma(t, pt_fma(t, 1.98412698e-04f, 1.38888889e-03f),
              8.33333333e-03f), 4.16666667e-02f), 1.66666667e-01f), 0.5f), 1.0f), 1.0f);
    int ki = (int)k;
    return p * as_float((uint32_t)(ki + 127) << 23);
}

PT_HD float pow_spec(float x, float y) { return exp2_spec(y * log2_spec(x)); }

// Color::FromSrgb (ShadingHelpers.hlsli:29)
PT_HD float from_srgb(float c)
{
    c = saturate(c);
    if (c <= 0.04045f) return c * (1.0f / 12.92f);
    return pow_spec((c + 0.055f) * (1.0f / 1.055f), 2.4f);
}

// Color::Luminance
PT_HD float luminance(f3 c) { return dot(c, make_f3(0.2126f, 0.7152f, 0.0722f)); }

// ---------------------------------------------------------------- Geometry
struct Basis {
    f3 T, B, N;
};

// Geometry::GetBasis (SurfaceVectors.hlsli:14)
PT_HD Basis get_basis(f3 N)
{
    float sz = sign(N.z);
    float a = pt_rcp(sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    Basis m;
    m.T = make_f3(c * N.x * a - 1.0f, sz * b, c);
    m.B = make_f3(b, N.y * ya - sz, N.y);
    m.N = N;
    return m;
}
PT_HD f3 rotate_vector(const Basis& m, f3 v) { return make_f3(dot(m.T, v), dot(m.B, v), dot(m.N, v)); }
PT_HD f3 rotate_vector_inverse(const Basis& m, f3 v)
{
    return make_f3(pt_fma(v.z, m.N.x, pt_fma(v.y, m.B.x, v.x * m.T.x)),
                   pt_fma(v.z, m.N.y, pt_fma(v.y, m.B.y, v.x * m.T.y)),
                   pt_fma(v.z, m.N.z, pt_fma(v.y, m.B.z, v.x * m.T.z)));
}
PT_HD f3 reflect(f3 i, f3 n) { float k = 2.0f * dot(n, i); return mad(-k, n, i); }
PT_HD f3 refract(f3 i, f3 n, float eta)
{
    float c = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - c * c);
    if (k < 0.0f) return make_f3(0.0f, 0.0f, 0.0f);
    float a = pt_fma(eta, c, pt_sqrt(k));
    return make_f3(pt_fma(-a, n.x, eta * i.x), pt_fma(-a, n.y, eta * i.y), pt_fma(-a, n.z, eta * i.z));
}

// ---------------------------------------------------------------- ImportanceSampling / BRDF terms
PT_HD f3 cosine_ray(float u0, float u1)
{
    float s, c;
    sincos_2pi(u0, s, c);
    float cos_t = sqrt01(u1);
    float sin_t = sqrt01(1.0f - cos_t * cos_t);
    return make_f3(sin_t * c, sin_t * s, cos_t);
}

PT_HD f3 vndf_ray(float u0, float u1, float roughness, f3 Vl)
{
    float m = roughness * roughness;
    f3 Vh = normalize(make_f3(m * Vl.x, m * Vl.y, Vl.z));
    float s, c;
    sincos_2pi(u0, s, c);
    float z = pt_fma(1.0f - u1, 1.0f + Vh.z, -Vh.z);
    float sr = sqrt01(1.0f - z * z);
    f3 Nh = make_f3(pt_fma(sr, c, Vh.x), pt_fma(sr, s, Vh.y), z + Vh.z);
    return normalize(make_f3(m * Nh.x, m * Nh.y, pt_max(Nh.z, 1e-7f)));
}

PT_HD float distribution_term(float roughness, float noh)
{
    // robust GGX form (DESIGN.md "Frozen arithmetic spec", deviation D1): finite at NoH == 1, roughness 2e-3
    float m = roughness * roughness;
    float m2 = m * m;
    float t = pt_fma(-(noh * noh), 0.99999994f - m2, 1.0f);
    float a = pt_max(m, 1e-6f) / t;
    return (a * a) * kInvPi;
}

PT_HD float vndf_pdf(f3 Vl, float noh, float roughness)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float nov = pt_abs(Vl.z);
    float d = distribution_term(roughness, noh);
    return d * pt_half_rcp(nov + pt_sqrt(pt_fma(1.0f - m2, nov * nov, m2)));  // D G1 / (4 NoV), one division (0.5 / x)
}

PT_HD float geometry_term_mod(float roughness, float nol, float nov)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float a = nov * sqrt01(pt_fma(pt_fma(-m2, nol, nol), nol, m2));
    float b = nol * sqrt01(pt_fma(pt_fma(-m2, nov, nov), nov, m2));
    return pt_half_rcp(a + b);  // 0.5 / (a + b)
}

PT_HD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }

PT_HD f3 fresnel_schlick(f3 f0, float voh)
{
    float p = pow5(1.0f - voh);
    return make_f3(pt_fma(1.0f - f0.x, p, f0.x), pt_fma(1.0f - f0.y, p, f0.y), pt_fma(1.0f - f0.z, p, f0.z));
}

PT_HD float fresnel_dielectric(float eta, float von)
{
    float sa2 = eta * eta * (1.0f - von * von);
    float ca = sqrt01(1.0f - sa2);
    float rs = (eta * von - ca) / (eta * von + ca);
    float rp = (eta * ca - von) / (eta * ca + von);
    return 0.5f * (rs * rs + rp * rp);
}

PT_HD float diffuse_term(float roughness, float nol, float nov, float voh)
{
    float f = pt_fma(2.0f * voh * voh, roughness, -0.5f);
    float fdv = pt_fma(f, pow5(1.0f - nov), 1.0f);
    float fdl = pt_fma(f, pow5(1.0f - nol), 1.0f);
    return fdv * fdl * kInvPi;
}

PT_HD f3 environment_term_rtg(f3 f0, float nov, float roughness)
{
    float m = roughness * roughness;
    float x1 = nov, x2 = nov * nov, x3 = nov * x2;
    float y1 = m, y3 = m * (m * m);
    float b_num = pt_fma(pt_fma(-0.755907f, x1, 1.29678f), y1, pt_fma(-1.28514f, x1, 0.99044f));
    float b_den = pt_fma(pt_fma(316.627f, x3, pt_fma(626.13f, x1, 121.563f)), y3,
                         pt_fma(pt_fma(222.592f, x3, pt_fma(-27.0302f, x1, 20.3225f)), y1,
                                pt_fma(59.4188f, x3, pt_fma(2.92338f, x1, 1.0f))));
    float s_num = pt_fma(pt_fma(-9.04756f, x1, 9.0632f), y1, pt_fma(3.32707f, x1, 0.0365463f));
    float s_den = pt_fma(pt_fma(-20.2123f, x3, pt_fma(19.7886f, x2, 5.56589f)), y3,
                         pt_fma(pt_fma(9.22949f, x3, pt_fma(-16.3174f, x2, 9.04401f)), y1,
                                pt_fma(-1.36772f, x3, pt_fma(3.59685f, x2, 1.0f))));
    float bias = b_num / b_den;
    float scale = s_num / s_den;
    return make_f3(saturate(pt_fma(f0.x, scale, bias)), saturate(pt_fma(f0.y, scale, bias)), saturate(pt_fma(f0.z, scale, bias)));
}

// ---------------------------------------------------------------- environment: GetEnvironmentLightColor (ShadingHelpers.hlsli:11-30)
// env_rgba = SceneData.EnvironmentLightColor; a >= 0: constant colour, else procedural sky.
PT_HD f3 environment_color(float er, float eg, float eb, float ea, f3 d)
{
    if (ea >= 0.0f) return make_f3(er, eg, eb);
    // Procedural sky FromSrgb(lerp(1, (0.5, 0.7, 1), (d.y + 1) / 2)): degree-7 polynomial fits in s = d.y of the exact
    // per-channel function (DESIGN.md spec S5); blue is exactly 1.
    float s = d.y;
    float r = pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(-3.77875438e-07f, s, -2.31609647e-06f), s, -1.62075557e-05f), s, -0.000163228658f), s, -0.00350578595f), s, 0.0846645609f), s, -0.389457047f), s, 0.522521555f);
    float g = pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(pt_fma(-5.82704285e-09f, s, -6.79562859e-08f), s, -9.30696501e-07f), s, -1.75486421e-05f), s, -0.000705873303f), s, 0.0319407657f), s, -0.275298983f), s, 0.69207108f);
    return make_f3(r, g, 1.0f);
}

// ---------------------------------------------------------------- ray-sphere (replaces CastRay's triangle hit, RaytracingHelpers.hlsli:57-133)
// Unit-length d.  Nearest root t with tmin < t < tmax.  Returns false on miss.
PT_HD bool intersect_sphere(f3 o, f3 d, float tmin, float tmax, f3 C, float r, float& t_out)
{
    f3 f = o - C;
    float bp = -dot(f, d);
    float r2 = r * r;
    float cc = dot(f, f) - r2;
    // Early out (changes no result for tmin >= 0): origin outside the sphere and the closest approach behind it ->
    // q = bp - sqrt(disc) < 0 and cc / q < 0, i.e. both roots are negative and the spec'd test t > tmin fails anyway.
    // It skips the sqrt + division for every sphere a ray is leaving (the ground sphere, for every bounce ray).
    if (cc > 0.0f && bp < 0.0f) return false;
    f3 l = mad(bp, d, f);
    float disc = r2 - dot(l, l);
    if (!(disc >= 0.0f)) return false;
    float sq = pt_sqrt(disc);
    float q = bp + (bp >= 0.0f ? sq : -sq);
    float ta = cc / q;
    float tb = q;
    float t0 = ta < tb ? ta : tb;
    float t1 = ta < tb ? tb : ta;
    float t = t0 > tmin ? t0 : t1;
    if (t > tmin && t < tmax) { t_out = t; return true; }
    return false;
}

struct HitFrame {
    f3 P;          // hit position re-projected onto the sphere
    f3 N;          // outward geometric normal (= flat = geometric normal of HitInfo.hlsli)
    float offset;  // PositionOffset analogue
    bool front;    // IsFrontFace = dot(N, dir) < 0 (HitInfo.hlsli:47,60)
};

PT_HD HitFrame hit_frame(f3 o, f3 d, float t, f3 C, float r)
{
    HitFrame h;
    f3 P0 = mad(t, d, o);
    h.N = normalize(P0 - C);
    h.P = mad(r, h.N, C);
    float mx = pt_max(pt_max(pt_abs(h.P.x), pt_abs(h.P.y)), pt_max(pt_abs(h.P.z), r));
    h.offset = kOffsetScale * mx;
    h.front = dot(h.N, d) < 0.0f;
    return h;
}

// HitInfo::GetSafeWorldRayOrigin (HitInfo.hlsli:96-99)
PT_HD f3 spawn_origin(f3 P, f3 N, float offset, f3 L)
{
    float sg = sign(dot(L, N));
    return mad(offset, N * sg, P);
}

}  // namespace pt
