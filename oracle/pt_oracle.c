/* pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See pt_oracle.h.
 *
 * PARITY UNPINNED (no reference tests / golden vectors exist; MathLib is absent).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (see oracle/Makefile).  Every floating-point
 * operation below is written in the exact order of DESIGN.md "Frozen arithmetic spec";
 * fused multiply-adds appear ONLY where spelled FMA(); min/max/saturate are explicit
 * ternaries so NaN behaviour is defined.  Citations are into /root/reference.
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* s*a + b, one fma per component */
static inline v3 v_mad(float s, v3 a, v3 b) { return V3(FMA(s, a.x, b.x), FMA(s, a.y, b.y), FMA(s, a.z, b.z)); }
/* spec dot: fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)) */
static inline float v_dot(v3 a, v3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }
/* spec normalize: v * (1/sqrt(dot(v,v))) */
static inline v3 v_normalize(v3 a) { float inv = 1.0f / sqrtf(v_dot(a, a)); return v_scale(a, inv); }
static inline float f_abs(float x) { return fabsf(x); }
static inline float f_max(float a, float b) { return a > b ? a : b; }
static inline float f_min(float a, float b) { return a < b ? a : b; }
/* HLSL saturate: NaN -> 0 */
static inline float f_sat(float x) { return !(x > 0.0f) ? 0.0f : (x > 1.0f ? 1.0f : x); }
static inline float f_sqrt01(float x) { return sqrtf(f_sat(x)); }       /* Math::Sqrt01 */
static inline float f_sign(float x) { return x >= 0.0f ? 1.0f : -1.0f; } /* Math::Sign, Sign(0)=+1 */
static inline int f_finite(float x) { return isfinite(x); }
static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#define PT_PI 3.14159265358979323846f
#define PT_INV_PI (1.0f / PT_PI)
#define PT_OFFSET_SCALE 1.52587890625e-05f /* 2^-16, build-defined sphere spawn offset */
#define PT_MIN_ROUGHNESS 2e-3f             /* BxDF.hlsli:19 */

/* ------------------------------------------------------------------ RNG (Appendix A) */
uint32_t oracle_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static inline uint32_t hash_combine(uint32_t seed, uint32_t v)
{
    return seed ^ (oracle_hash(v) + 0x9E3779B9u + (seed << 6) + (seed >> 2));
}
/* Rng::Hash::Initialize(pixel, frame): Raytracing.hlsl:108 */
uint32_t oracle_rng_init(uint32_t px, uint32_t py, uint32_t frame)
{
    return hash_combine(oracle_hash(frame + 0x035F9F29u), (px << 16) | py);
}
uint32_t oracle_rng_next(uint32_t *state) { *state = oracle_hash(*state); return *state; }
/* GetFloat = 2 - asfloat((u>>9)|0x3F800000), range (0,1] */
float oracle_rng_float(uint32_t *state)
{
    uint32_t u = oracle_rng_next(state);
    return 2.0f - as_float((u >> 9) | 0x3F800000u);
}

/* ------------------------------------------------------------------ Halton (HaltonSampler.ixx:32-34) */
float oracle_halton(uint32_t index, uint32_t base)
{
    if (base == 2) { /* bit reversal * 2^-32 */
        uint32_t v = index;
        v = (v << 16) | (v >> 16);
        v = ((v & 0x00FF00FFu) << 8) | ((v & 0xFF00FF00u) >> 8);
        v = ((v & 0x0F0F0F0Fu) << 4) | ((v & 0xF0F0F0F0u) >> 4);
        v = ((v & 0x33333333u) << 2) | ((v & 0xCCCCCCCCu) >> 2);
        v = ((v & 0x55555555u) << 1) | ((v & 0xAAAAAAAAu) >> 1);
        return (float)v * 2.3283064365386963e-10f;
    }
    float f = 1.0f, r = 0.0f, fb = (float)base;
    uint32_t i = index;
    while (i > 0) {
        f = f / fb;
        r = r + f * (float)(i % base);
        i = i / base;
    }
    return r;
}

/* ------------------------------------------------------------------ build-defined sincos / log2 / exp2 / pow */
/* sin, cos of 2*pi*u, u in [0,1].  Quadrant reduction on u (exact), then odd/even polynomials
 * on [-pi/4, pi/4] evaluated with fma Horner chains. */
void oracle_sincos_2pi(float u, float *so, float *co)
{
    float x = u * 4.0f;
    float k = floorf(x + 0.5f);
    float y = x - k;                       /* [-0.5, 0.5], exact */
    float z = y * 1.57079632679489661923f; /* pi/2 */
    float z2 = z * z;
    float ps = FMA(z2, FMA(z2, FMA(z2, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
    float s = FMA(z * z2, ps, z);
    float pc = FMA(z2, FMA(z2, FMA(z2, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
    float c = FMA(z2 * z2, pc, FMA(z2, -0.5f, 1.0f));
    int q = ((int)k) & 3;
    float rs, rc;
    if (q == 0) { rs = s; rc = c; }
    else if (q == 1) { rs = c; rc = -s; }
    else if (q == 2) { rs = -s; rc = -c; }
    else { rs = -c; rc = s; }
    *so = rs; *co = rc;
}

/* log2 for finite x > 0 (normal numbers) */
float oracle_log2(float x)
{
    uint32_t b = as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = as_float((b & 0x007FFFFFu) | 0x3F800000u); /* [1,2) */
    if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = FMA(s2, FMA(s2, FMA(s2, FMA(s2, 0.111111111f, 0.142857143f), 0.2f), 0.333333333f), 1.0f);
    float ln_m = (2.0f * s) * p;
    return FMA(ln_m, 1.44269504088896340736f, (float)e);
}

/* 2^y for |y| < 126 */
float oracle_exp2(float y)
{
    float k = floorf(y + 0.5f);
    float f = y - k; /* [-0.5, 0.5] */
    float t = f * 0.69314718055994530942f;
    float p = FMA(t, FMA(t, FMA(t, FMA(t, FMA(t, FMA(t, FMA(t, 1.98412698e-04f, 1.38888889e-03f), 8.33333333e-03f),
                  4.16666667e-02f), 1.66666667e-01f), 0.5f), 1.0f), 1.0f);
    int ki = (int)k;
    return p * as_float((uint32_t)(ki + 127) << 23);
}

float oracle_pow(float x, float y) { return oracle_exp2(y * oracle_log2(x)); }

/* Color::FromSrgb, one channel (c saturated) */
float oracle_from_srgb(float c)
{
    c = f_sat(c);
    if (c <= 0.04045f) return c * (1.0f / 12.92f);
    return oracle_pow((c + 0.055f) * (1.0f / 1.055f), 2.4f);
}

static inline float luminance(v3 c) { return v_dot(c, V3(0.2126f, 0.7152f, 0.0722f)); }

/* ------------------------------------------------------------------ Geometry (Appendix A) */
typedef struct { v3 T, B, N; } basis3;

/* Geometry::GetBasis: SurfaceVectors.hlsli:14 */
static basis3 get_basis(v3 N)
{
    float sz = f_sign(N.z);
    float a = 1.0f / (sz + N.z);
    float ya = N.y * a;
    float b = N.x * ya;
    float c = N.x * sz;
    basis3 m;
    m.T = V3(c * N.x * a - 1.0f, sz * b, c);
    m.B = V3(b, N.y * ya - sz, N.y);
    m.N = N;
    return m;
}
void oracle_get_basis(const float n[3], float t[3], float b[3])
{
    basis3 m = get_basis(V3(n[0], n[1], n[2]));
    t[0] = m.T.x; t[1] = m.T.y; t[2] = m.T.z;
    b[0] = m.B.x; b[1] = m.B.y; b[2] = m.B.z;
}
/* world -> local */
static inline v3 rotate_vector(basis3 m, v3 v) { return V3(v_dot(m.T, v), v_dot(m.B, v), v_dot(m.N, v)); }
/* local -> world: v.x*T + v.y*B + v.z*N */
static inline v3 rotate_vector_inverse(basis3 m, v3 v)
{
    return V3(FMA(v.z, m.N.x, FMA(v.y, m.B.x, v.x * m.T.x)),
              FMA(v.z, m.N.y, FMA(v.y, m.B.y, v.x * m.T.y)),
              FMA(v.z, m.N.z, FMA(v.y, m.B.z, v.x * m.T.z)));
}
/* HLSL reflect(i,n) = i - 2*dot(n,i)*n */
static inline v3 reflect3(v3 i, v3 n) { float k = 2.0f * v_dot(n, i); return v_mad(-k, n, i); }
/* HLSL refract(i,n,eta) */
static inline v3 refract3(v3 i, v3 n, float eta)
{
    float c = v_dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - c * c);
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    float a = FMA(eta, c, sqrtf(k));
    return V3(FMA(-a, n.x, eta * i.x), FMA(-a, n.y, eta * i.y), FMA(-a, n.z, eta * i.z));
}

/* ------------------------------------------------------------------ sampling + BRDF terms (Appendix A) */
static v3 cosine_ray(float u0, float u1)
{
    float s, c;
    oracle_sincos_2pi(u0, &s, &c);
    float cos_t = f_sqrt01(u1);
    float sin_t = f_sqrt01(1.0f - cos_t * cos_t);
    return V3(sin_t * c, sin_t * s, cos_t);
}
void oracle_cosine_ray(const float u[2], float out[3])
{
    v3 r = cosine_ray(u[0], u[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* ImportanceSampling::VNDF::GetRay (spherical caps) */
static v3 vndf_ray(float u0, float u1, float roughness, v3 Vl)
{
    float m = roughness * roughness;
    v3 Vh = v_normalize(V3(m * Vl.x, m * Vl.y, Vl.z));
    float s, c;
    oracle_sincos_2pi(u0, &s, &c);
    float z = FMA(1.0f - u1, 1.0f + Vh.z, -Vh.z);
    float sr = f_sqrt01(1.0f - z * z);
    v3 Nh = V3(FMA(sr, c, Vh.x), FMA(sr, s, Vh.y), z + Vh.z);
    return v_normalize(V3(m * Nh.x, m * Nh.y, f_max(Nh.z, 1e-7f)));
}
void oracle_vndf_ray(const float u[2], float roughness, const float vl[3], float out[3])
{
    v3 r = vndf_ray(u[0], u[1], roughness, V3(vl[0], vl[1], vl[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* BRDF::DistributionTerm (GGX) */
float oracle_distribution_term(float roughness, float noh)
{
    /* robust form (upstream MathLib's default branch): the textbook ((NoH*m2 - NoH)*NoH + 1) form of SURVEY
     * Appendix A evaluates to 0 when NoH rounds to 1 at roughness 2e-3 -> D = inf -> f/pdf = NaN on the demo's
     * roughness-0 ground mirror (DESIGN.md "Frozen arithmetic spec", deviation D1). */
    float m = roughness * roughness;
    float m2 = m * m;
    float t = FMA(-(noh * noh), 0.99999994f - m2, 1.0f);
    float a = f_max(m, 1e-6f) / t;
    return (a * a) * PT_INV_PI;
}
/* VNDF::GetPDF(Vlocal, NoH, roughness) = D * G1(|Vl.z|) / (4 |Vl.z|) */
float oracle_vndf_pdf(const float vl[3], float noh, float roughness)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float nov = f_abs(vl[2]);
    /* D * G1 / (4 NoV) with G1 = 2 NoV / (NoV + sqrt(m2 + (1 - m2) NoV^2)), simplified to one division */
    float d = oracle_distribution_term(roughness, noh);
    return d * (0.5f / (nov + sqrtf(FMA(1.0f - m2, nov * nov, m2))));
}
/* BRDF::GeometryTermMod (Smith height-correlated / (4 NoL NoV)) */
float oracle_geometry_term_mod(float roughness, float nol, float nov)
{
    float m = roughness * roughness;
    float m2 = m * m;
    float a = nov * f_sqrt01(FMA(FMA(-m2, nol, nol), nol, m2));
    float b = nol * f_sqrt01(FMA(FMA(-m2, nov, nov), nov, m2));
    return 0.5f / (a + b);
}
static inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
/* BRDF::FresnelTerm (Schlick): F0 + (1-F0)(1-VoH)^5 */
static inline v3 fresnel_schlick(v3 f0, float voh)
{
    float p = pow5(1.0f - voh);
    return V3(FMA(1.0f - f0.x, p, f0.x), FMA(1.0f - f0.y, p, f0.y), FMA(1.0f - f0.z, p, f0.z));
}
/* BRDF::FresnelTerm_Dielectric(eta, VoN) */
float oracle_fresnel_dielectric(float eta, float von)
{
    float sa2 = eta * eta * (1.0f - von * von);
    float ca = f_sqrt01(1.0f - sa2);
    float rs = (eta * von - ca) / (eta * von + ca);
    float rp = (eta * ca - von) / (eta * ca + von);
    return 0.5f * (rs * rs + rp * rp);
}
/* BRDF::DiffuseTerm (Burley) */
float oracle_diffuse_term(float roughness, float nol, float nov, float voh)
{
    float f = FMA(2.0f * voh * voh, roughness, -0.5f);
    float fdv = FMA(f, pow5(1.0f - nov), 1.0f);
    float fdl = FMA(f, pow5(1.0f - nol), 1.0f);
    return fdv * fdl * PT_INV_PI;
}
/* BRDF::EnvironmentTerm_Rtg (RT Gems ch.32 fit) */
static v3 environment_term_rtg(v3 f0, float nov, float roughness)
{
    float m = roughness * roughness;
    float x1 = nov, x2 = nov * nov, x3 = nov * x2;
    float y1 = m, y2 = m * m, y3 = m * y2;
    /* mul(M, X) rows dotted with Y, all as left-to-right fma chains */
    float b_num = FMA(FMA(-0.755907f, x1, 1.29678f), y1, FMA(-1.28514f, x1, 0.99044f));
    float b_den = FMA(FMA(316.627f, x3, FMA(626.13f, x1, 121.563f)), y3,
                      FMA(FMA(222.592f, x3, FMA(-27.0302f, x1, 20.3225f)), y1,
                          FMA(59.4188f, x3, FMA(2.92338f, x1, 1.0f))));
    float s_num = FMA(FMA(-9.04756f, x1, 9.0632f), y1, FMA(3.32707f, x1, 0.0365463f));
    float s_den = FMA(FMA(-20.2123f, x3, FMA(19.7886f, x2, 5.56589f)), y3,
                      FMA(FMA(9.22949f, x3, FMA(-16.3174f, x2, 9.04401f)), y1,
                          FMA(-1.36772f, x3, FMA(3.59685f, x2, 1.0f))));
    float bias = b_num / b_den;
    float scale = s_num / s_den;
    (void)y2;
    return V3(f_sat(FMA(f0.x, scale, bias)), f_sat(FMA(f0.y, scale, bias)), f_sat(FMA(f0.z, scale, bias)));
}
void oracle_environment_term_rtg(const float f0[3], float nov, float roughness, float out[3])
{
    v3 r = environment_term_rtg(V3(f0[0], f0[1], f0[2]), nov, roughness);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ environment (ShadingHelpers.hlsli:11-30) */
struct tex_ctx_s;
static v3 environment_texture(const struct tex_ctx_s *tc, const PtSceneData *sd, v3 d);
static v3 environment_color(const struct tex_ctx_s *tc, const PtSceneData *sd, v3 d)
{
    if (sd->EnvironmentLightTextureDescriptor != 0xFFFFFFFFu && tc) return environment_texture(tc, sd, d); /* :13-24; oracle_render_textured has checked the descriptor */
    if (sd->EnvironmentLightColor[3] >= 0.0f)
        return V3(sd->EnvironmentLightColor[0], sd->EnvironmentLightColor[1], sd->EnvironmentLightColor[2]);
    /* Procedural sky: FromSrgb(lerp(1, (0.5, 0.7, 1), (d.y + 1) / 2)) per channel (ShadingHelpers.hlsli:29).  Each channel
     * is a smooth function of s = d.y on [-1, 1]; it is evaluated as the degree-7 polynomial fit of that exact function
     * (Chebyshev fit, |error| < 1e-9 in exact arithmetic, ~6e-8 in fp32 Horner) instead of three pow() calls.  Blue is
     * lerp(1, 1, t) = 1 and FromSrgb(1) = 1 exactly.  DESIGN.md "Frozen arithmetic spec" S5. */
    float s = d.y;
    float r = FMA(FMA(FMA(FMA(FMA(FMA(FMA(-3.77875438e-07f, s, -2.31609647e-06f), s, -1.62075557e-05f), s, -0.000163228658f), s, -0.00350578595f), s, 0.0846645609f), s, -0.389457047f), s, 0.522521555f);
    float g = FMA(FMA(FMA(FMA(FMA(FMA(FMA(-5.82704285e-09f, s, -6.79562859e-08f), s, -9.30696501e-07f), s, -1.75486421e-05f), s, -0.000705873303f), s, 0.0319407657f), s, -0.275298983f), s, 0.69207108f);
    return V3(r, g, 1.0f);
}
void oracle_sky(const PtSceneData *scene, const float dir[3], float out[3])
{
    v3 c = environment_color(NULL, scene, V3(dir[0], dir[1], dir[2])); out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* ------------------------------------------------------------------ ray-sphere (build-defined, replaces CastRay) */
/* Stable quadratic for a unit-length direction.  Nearest root t with tmin < t < tmax. */
static int intersect_sphere(v3 o, v3 d, float tmin, float tmax, const PtSphere *s, float *t_out)
{
    v3 f = v_sub(o, V3(s->cx, s->cy, s->cz));
    float bp = -v_dot(f, d);
    v3 l = v_mad(bp, d, f);
    float r2 = s->r * s->r;
    float disc = r2 - v_dot(l, l);
    if (!(disc >= 0.0f)) return 0;
    float sq = sqrtf(disc);
    float q = bp + (bp >= 0.0f ? sq : -sq);
    float cc = v_dot(f, f) - r2;
    float ta = cc / q;
    float tb = q;
    float t0 = ta < tb ? ta : tb;
    float t1 = ta < tb ? tb : ta;
    float t = t0 > tmin ? t0 : t1;
    if (t > tmin && t < tmax) { *t_out = t; return 1; }
    return 0;
}
int oracle_intersect_sphere(const float o[3], const float d[3], float tmin, float tmax, const PtSphere *s, float *t)
{
    return intersect_sphere(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmin, tmax, s, t);
}

typedef struct {
    int hit;
    uint32_t id;
    float t;
    v3 P, N;      /* re-projected position, outward geometric normal */
    float offset; /* spawn offset */
    int front;
    v3 shadingN;  /* N or -N (HitInfo.hlsli:60-64) */
} hit_t;

/* ------------------------------------------------------------------------------------------------------------------
 * Closest hit.  Brute force over all spheres is the definition (ties -> lowest id: strict < in id order).  For scenes of
 * more than OBVH_MIN_SPHERES spheres the oracle builds its own acceleration structure -- a median-split BVH with padded
 * boxes and a double-precision slab test, unrelated to the product's LBVH -- that returns exactly the brute-force answer
 * (tests/test_oracle_kat.py compares the two); it makes the 2^20-sphere configuration runnable on the CPU (BASELINE.md
 * section 2: "CPU-BVH (C5)").
 * ---------------------------------------------------------------------------------------------------------------- */
#define OBVH_MIN_SPHERES 64u
#define OBVH_LEAF 4u

typedef struct {
    float lo[3], hi[3];
    uint32_t left, right; /* children (internal) */
    uint32_t first, count; /* leaf: prim[first .. first + count) ; count == 0 for internal nodes */
} onode;

typedef struct {
    onode *nodes;
    uint32_t *prim;
    uint32_t n_nodes;
} obvh;

static void obvh_bounds(const PtSphere *sph, const uint32_t *prim, uint32_t first, uint32_t count, float pad, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (uint32_t k = first; k < first + count; k++) {
        const PtSphere *s = &sph[prim[k]];
        const float c[3] = { s->cx, s->cy, s->cz };
        for (int a = 0; a < 3; a++) {
            lo[a] = f_min(lo[a], c[a] - s->r - pad);
            hi[a] = f_max(hi[a], c[a] + s->r + pad);
        }
    }
}

static float obvh_key(const PtSphere *s, int axis) { return axis == 0 ? s->cx : (axis == 1 ? s->cy : s->cz); }

/* quickselect: afterwards prim[first .. mid) have centroid[axis] <= those of prim[mid .. last) */
static void obvh_select(const PtSphere *sph, uint32_t *prim, uint32_t first, uint32_t last, uint32_t mid, int axis)
{
    uint32_t lo = first, hi = last; /* [lo, hi) */
    uint32_t seed = 12345u;
    while (hi - lo > 1) {
        seed = seed * 1664525u + 1013904223u;
        const float pivot = obvh_key(&sph[prim[lo + seed % (hi - lo)]], axis);
        uint32_t i = lo, j = lo, k = hi; /* three-way partition: [lo,i) < pivot, [i,j) == pivot, [k,hi) > pivot */
        while (j < k) {
            const float v = obvh_key(&sph[prim[j]], axis);
            if (v < pivot) { uint32_t t = prim[i]; prim[i] = prim[j]; prim[j] = t; i++; j++; }
            else if (v > pivot) { k--; uint32_t t = prim[j]; prim[j] = prim[k]; prim[k] = t; }
            else j++;
        }
        if (mid < i) hi = i;
        else if (mid >= k) lo = k;
        else return; /* mid falls in the run of equal keys */
    }
}

static obvh *obvh_build(const PtSphere *sph, uint32_t n)
{
    obvh *b = (obvh *)calloc(1, sizeof(obvh));
    b->prim = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    b->nodes = (onode *)malloc((size_t)(2u * n) * sizeof(onode));
    for (uint32_t i = 0; i < n; i++) b->prim[i] = i;
    /* padding: 2^-15 of the largest coordinate magnitude (4x the product's, so the structure is conservative in its own right) */
    float smax = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        const PtSphere *s = &sph[i];
        smax = f_max(smax, f_max(f_max(f_abs(s->cx) + s->r, f_abs(s->cy) + s->r), f_abs(s->cz) + s->r));
    }
    const float pad = smax * 3.0517578125e-05f;
    uint32_t *stack = (uint32_t *)malloc((size_t)(2u * n + 2u) * sizeof(uint32_t));
    uint32_t sp = 0;
    b->n_nodes = 1;
    b->nodes[0].first = 0; b->nodes[0].count = n;
    stack[sp++] = 0;
    while (sp) {
        onode *nd = &b->nodes[stack[--sp]];
        obvh_bounds(sph, b->prim, nd->first, nd->count, pad, nd->lo, nd->hi);
        if (nd->count <= OBVH_LEAF) continue;
        /* split at the median of the centroids along their widest axis */
        float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t k = nd->first; k < nd->first + nd->count; k++)
            for (int a = 0; a < 3; a++) {
                const float v = obvh_key(&sph[b->prim[k]], a);
                clo[a] = f_min(clo[a], v); chi[a] = f_max(chi[a], v);
            }
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const uint32_t mid = nd->first + nd->count / 2u;
        obvh_select(sph, b->prim, nd->first, nd->first + nd->count, mid, axis);
        const uint32_t l = b->n_nodes++, r = b->n_nodes++;
        b->nodes[l].first = nd->first; b->nodes[l].count = mid - nd->first;
        b->nodes[r].first = mid; b->nodes[r].count = nd->first + nd->count - mid;
        nd->left = l; nd->right = r; nd->count = 0;
        stack[sp++] = l; stack[sp++] = r;
    }
    free(stack);
    return b;
}

static void obvh_free(obvh *b)
{
    if (!b) return;
    free(b->nodes); free(b->prim); free(b);
}

/* conservative ray / box overlap on [tmin, tmax] in double precision (the boxes are padded far beyond its rounding) */
static int obvh_hits_box(const onode *nd, const double o[3], const double inv[3], double tmin, double tmax)
{
    double t0 = tmin, t1 = tmax;
    for (int a = 0; a < 3; a++) {
        double ta = ((double)nd->lo[a] - o[a]) * inv[a], tb = ((double)nd->hi[a] - o[a]) * inv[a];
        if (ta != ta || tb != tb) continue; /* 0 * inf: the origin lies on a plane of a slab the ray is parallel to */
        if (ta > tb) { double t = ta; ta = tb; tb = t; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    return t0 <= t1 * (1.0 + 1e-12) + 1e-300;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Alpha-tested hits (spec S10).  The reference flags the geometry of an object whose AlphaMode is not Opaque as non-opaque
 * (Source/Scene.ixx:242-243); TraceRay then commits a candidate of such an object only if IsOpaque says so
 * (Shaders/RaytracingHelpers.hlsli:19-43): alpha = BaseColor.a, times the alpha of the base-colour map at the candidate's
 * texture coordinates when EvaluateBaseColor's condition holds (any(BaseColor > 0) and a map, Shaders/ShadingHelpers.hlsli:61-72),
 * accepted iff alpha >= AlphaCutoff (:105-115) -- Mask and Blend alike.  A candidate there is a triangle; here it is one
 * crossing of the ray with the sphere's surface: the near root of the quadratic, then the far one (the back of the sphere seen
 * from inside), each tested at its own texture coordinates.  "The next crossing" = intersect_sphere with tmin = the rejected t.
 * ---------------------------------------------------------------------------------------------------------------- */
struct tex_ctx_s;
typedef struct { const PtMaterial *mat; const struct tex_ctx_s *tc; } alpha_ctx;
static int crossing_is_opaque(const alpha_ctx *a, const PtSphere *s, uint32_t id, v3 o, v3 d, float t);
static size_t sizeof_tex_ctx(void);
static void tex_ctx_init_opaque(struct tex_ctx_s *c, const OracleTextures *t);

/* the first crossing of sphere `id` in (tmin, tmax) that the alpha test accepts; a == NULL: every sphere is opaque */
static int sphere_candidate(const alpha_ctx *a, v3 o, v3 d, float tmin, float tmax, const PtSphere *s, uint32_t id, float *t_out)
{
    float tm = tmin, t;
    while (intersect_sphere(o, d, tm, tmax, s, &t)) {
        if (!a || a->mat[id].AlphaMode == PT_ALPHA_OPAQUE || crossing_is_opaque(a, s, id, o, d, t)) { *t_out = t; return 1; }
        tm = t; /* rejected: on to the crossing behind it */
    }
    return 0;
}

static void closest_hit_id(const obvh *b, const PtSphere *sph, uint32_t n, v3 o, v3 d, float tmin, float tmax, float *best_t, uint32_t *best_id, const alpha_ctx *alpha)
{
    float best = tmax;
    uint32_t id = 0xFFFFFFFFu;
    if (!b) {
        for (uint32_t i = 0; i < n; i++) {
            float t;
            if (sphere_candidate(alpha, o, d, tmin, best, &sph[i], i, &t)) { best = t; id = i; }
        }
    } else {
        const double oo[3] = { o.x, o.y, o.z };
        const double inv[3] = { 1.0 / (double)d.x, 1.0 / (double)d.y, 1.0 / (double)d.z };
        uint32_t stack[128];
        uint32_t sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const onode *nd = &b->nodes[stack[--sp]];
            if (!obvh_hits_box(nd, oo, inv, (double)tmin, (double)best)) continue;
            if (nd->count) {
                for (uint32_t k = nd->first; k < nd->first + nd->count; k++) {
                    const uint32_t i = b->prim[k];
                    float t;
                    /* same acceptance as the brute-force loop: inside (tmin, tmax); nearer wins, ties go to the lower id */
                    if (sphere_candidate(alpha, o, d, tmin, tmax, &sph[i], i, &t) && (t < best || (t == best && id != 0xFFFFFFFFu && i < id))) { best = t; id = i; }
                }
            } else if (sp + 2 <= 128) {
                stack[sp++] = nd->left; stack[sp++] = nd->right;
            }
        }
    }
    *best_t = best; *best_id = id;
}

static void cast_ray(const void *accel, const PtSphere *sph, uint32_t n, v3 o, v3 d, float tmin, float tmax, hit_t *h, const alpha_ctx *alpha)
{
    float best;
    uint32_t best_id;
    closest_hit_id((const obvh *)accel, sph, n, o, d, tmin, tmax, &best, &best_id, alpha);
    h->hit = best_id != 0xFFFFFFFFu;
    h->id = best_id;
    h->t = best;
    if (h->hit) {
        const PtSphere *s = &sph[best_id];
        v3 C = V3(s->cx, s->cy, s->cz);
        v3 P0 = v_mad(best, d, o);
        v3 N = v_normalize(v_sub(P0, C));
        v3 P = v_mad(s->r, N, C);
        float mx = f_max(f_max(f_abs(P.x), f_abs(P.y)), f_max(f_abs(P.z), s->r));
        h->P = P; h->N = N;
        h->offset = PT_OFFSET_SCALE * mx;
        h->front = v_dot(N, d) < 0.0f;
        h->shadingN = h->front ? N : v_neg(N);
    }
}

/* test hook: closest hit of one ray through the brute-force loop (use_bvh = 0) or the oracle's BVH */
int oracle_closest_hit(const PtSphere *spheres, uint32_t n, const float o[3], const float d[3], float tmin, float tmax, int use_bvh,
                       void **bvh_cache, float *t, uint32_t *id)
{
    obvh *b = NULL;
    if (use_bvh) {
        if (bvh_cache && *bvh_cache) b = (obvh *)*bvh_cache;
        else { b = obvh_build(spheres, n); if (bvh_cache) *bvh_cache = b; }
    }
    closest_hit_id(b, spheres, n, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmin, tmax, t, id, NULL);
    if (use_bvh && !bvh_cache) obvh_free(b);
    return *id != 0xFFFFFFFFu;
}
void oracle_free_bvh(void *bvh) { obvh_free((obvh *)bvh); }

/* the same query with alpha-tested hits (spec S10): materials carry AlphaMode / AlphaCutoff / BaseColor, textures (may be NULL)
 * the base-colour maps and rotations.  Brute force only (the definition). */
int oracle_closest_hit_alpha(const PtSphere *spheres, const PtMaterial *materials, uint32_t n, const OracleTextures *textures,
                             const float o[3], const float d[3], float tmin, float tmax, float *t, uint32_t *id)
{
    struct tex_ctx_s *tc = NULL;
    if (textures && textures->n_textures > 0) { tc = (struct tex_ctx_s *)malloc(sizeof_tex_ctx()); tex_ctx_init_opaque(tc, textures); }
    const alpha_ctx a = { materials, tc };
    closest_hit_id(NULL, spheres, n, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmin, tmax, t, id, &a);
    free(tc);
    return *id != 0xFFFFFFFFu;
}

void oracle_hit_frame(const float o[3], const float d[3], float t, const PtSphere *s,
                      float P[3], float N[3], float *offset, int *front)
{
    v3 oo = V3(o[0], o[1], o[2]), dd = V3(d[0], d[1], d[2]);
    v3 C = V3(s->cx, s->cy, s->cz);
    v3 P0 = v_mad(t, dd, oo);
    v3 n = v_normalize(v_sub(P0, C));
    v3 p = v_mad(s->r, n, C);
    float mx = f_max(f_max(f_abs(p.x), f_abs(p.y)), f_max(f_abs(p.z), s->r));
    P[0] = p.x; P[1] = p.y; P[2] = p.z; N[0] = n.x; N[1] = n.y; N[2] = n.z;
    *offset = PT_OFFSET_SCALE * mx;
    *front = v_dot(n, dd) < 0.0f;
}
/* HitInfo::GetSafeWorldRayOrigin (HitInfo.hlsli:96-99) + OffsetSpawnPoint (SelfIntersectionAvoidance.hlsli:113-117) */
static inline v3 spawn_origin(v3 P, v3 N, float offset, v3 L)
{
    float sg = f_sign(v_dot(L, N));
    return v_mad(offset, v_scale(N, sg), P);
}
void oracle_spawn_origin(const float P[3], const float N[3], float offset, const float L[3], float out[3])
{
    v3 r = spawn_origin(V3(P[0], P[1], P[2]), V3(N[0], N[1], N[2]), offset, V3(L[0], L[1], L[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ camera (Camera.hlsli:27-41, Math.hlsli:7-15) */
static void primary_ray(const PtCamera *cam, uint32_t px, uint32_t py, uint32_t w, uint32_t h,
                        v3 *o, v3 *d, float *tmin, float *tmax)
{
    float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;
    float u = ((float)px + 0.5f + cam->Jitter[0]) * inv_w;
    float v = ((float)py + 0.5f + cam->Jitter[1]) * inv_h;
    float nx = FMA(u, 2.0f, -1.0f);
    float ny = FMA(v, -2.0f, 1.0f);
    v3 R = V3(cam->RightDirection[0], cam->RightDirection[1], cam->RightDirection[2]);
    v3 U = V3(cam->UpDirection[0], cam->UpDirection[1], cam->UpDirection[2]);
    v3 F = V3(cam->ForwardDirection[0], cam->ForwardDirection[1], cam->ForwardDirection[2]);
    v3 dir = v_add(v_mad(ny, U, v_scale(R, nx)), F);
    dir = v_normalize(dir);
    float inv_cos = 1.0f / v_dot(v_normalize(F), dir);
    *o = V3(cam->Position[0], cam->Position[1], cam->Position[2]);
    *d = dir;
    *tmin = cam->NearDepth * inv_cos;
    *tmax = cam->FarDepth * inv_cos;
}
void oracle_primary_ray(const PtCamera *cam, uint32_t px, uint32_t py, uint32_t w, uint32_t h,
                        float o[3], float d[3], float *tmin, float *tmax)
{
    v3 oo, dd;
    primary_ray(cam, px, py, w, h, &oo, &dd, tmin, tmax);
    o[0] = oo.x; o[1] = oo.y; o[2] = oo.z; d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
}

/* ------------------------------------------------------------------ BSDF (BxDF.hlsli) */
enum { LOBE_DIFFUSE = 0, LOBE_SPECULAR = 1, LOBE_TRANSMISSION = 2 };

typedef struct {
    v3 BaseColor; float Metallic; v3 Albedo; float Roughness, IORi, IORo; v3 F0; float Transmission;
} bsdf_t;

/* BSDFSample::Initialize, BxDF.hlsli:45-66 (pow(x,2) restated as x*x) */
static void bsdf_init(bsdf_t *b, v3 base, float metallic, float roughness, float ior, float transmission, int front)
{
    b->BaseColor = base;
    b->Metallic = metallic;
    b->Albedo = v_scale(base, 1.0f - metallic);
    b->Roughness = f_max(PT_MIN_ROUGHNESS, roughness);
    b->IORi = 1.0f; b->IORo = ior;
    if (!front) { b->IORi = ior; b->IORo = 1.0f; }
    float r = (b->IORi - b->IORo) / (b->IORi + b->IORo);
    float f0d = r * r;
    /* lerp(f0d, baseColor, metallic) */
    b->F0 = V3(FMA(metallic, base.x - f0d, f0d), FMA(metallic, base.y - f0d, f0d), FMA(metallic, base.z - f0d, f0d));
    b->Transmission = transmission;
}

typedef struct { v3 FrontNg, Ns; basis3 basis; } surf_t;

/* SurfaceVectors::Initialize, SurfaceVectors.hlsli:10-15 */
static void surf_init(surf_t *s, int front, v3 Ng, v3 Ns)
{
    s->FrontNg = front ? Ng : v_neg(Ng);
    s->Ns = Ns;
    s->basis = get_basis(Ns);
}

/* EstimateDiffuseProbability, BxDF.hlsli:21-34 */
static float estimate_diffuse_probability(v3 albedo, v3 f0, float roughness, float nov)
{
    v3 fe = environment_term_rtg(f0, nov, roughness);
    float diffuse = luminance(v_mul(albedo, V3(1.0f - fe.x, 1.0f - fe.y, 1.0f - fe.z)));
    float specular = luminance(fe);
    float sum = diffuse + specular;
    float p = sum > 0.0f ? diffuse / sum : 1.0f;
    if (0.0f < p && p < 1.0f) { /* clamp(p, 0.05, 0.95) */
        p = p < 0.05f ? 0.05f : (p > 0.95f ? 0.95f : p);
    }
    return p;
}

/* ComputeLobeWeights, BxDF.hlsli:184-196 */
static void lobe_weights(const bsdf_t *b, const surf_t *s, v3 V, float w[3])
{
    float nov = f_abs(v_dot(s->Ns, V));
    float wt = b->Transmission * (1.0f - b->Metallic);
    float wr = 1.0f - wt;
    float pd = estimate_diffuse_probability(b->Albedo, b->F0, b->Roughness, nov);
    float ps = 1.0f - pd;
    w[LOBE_DIFFUSE] = pd * wr;
    w[LOBE_SPECULAR] = ps * wr;
    w[LOBE_TRANSMISSION] = wt;
}

/* FindLobe, BxDF.hlsli:198-212 */
static int find_lobe(const float w[3], float rnd)
{
    float weight = 0.0f;
    weight += w[2];
    if (rnd < weight) return 2;
    weight += w[1];
    if (rnd < weight) return 1;
    return 0;
}

/* Sample, BxDF.hlsli:214-226 and the three Sample* functions :81-86,110-118,148-170 */
static int bsdf_sample(const bsdf_t *b, const surf_t *s, v3 V, const float w[3], const float rnd[4], v3 *L, int *lobe)
{
    *lobe = find_lobe(w, rnd[0]);
    if (*lobe == LOBE_DIFFUSE) {
        *L = rotate_vector_inverse(s->basis, cosine_ray(rnd[1], rnd[2]));
        return v_dot(s->FrontNg, *L) > 0.0f;
    }
    v3 Vl = rotate_vector(s->basis, V);
    v3 H = rotate_vector_inverse(s->basis, vndf_ray(rnd[1], rnd[2], b->Roughness, Vl));
    if (*lobe == LOBE_SPECULAR) {
        *L = reflect3(v_neg(V), H);
        return v_dot(s->FrontNg, *L) > 0.0f;
    }
    float voh = f_abs(v_dot(V, H));
    float eta = b->IORi / b->IORo;
    if (eta * eta * (1.0f - voh * voh) > 1.0f || rnd[3] < oracle_fresnel_dielectric(eta, voh)) {
        *L = reflect3(v_neg(V), H);
    } else {
        *L = refract3(v_neg(V), H, eta);
        if (!f_finite(L->x) || !f_finite(L->y) || !f_finite(L->z)) *L = v_neg(V);
    }
    return 1;
}

/* ComputeHalfVector, BxDF.hlsli:228-245 */
static v3 half_vector(const bsdf_t *b, const surf_t *s, v3 L, v3 V, int transmissive)
{
    v3 N = s->FrontNg;
    v3 H;
    if (transmissive && v_dot(N, L) < 0.0f) {
        H = v_normalize(v_mad(b->IORo, L, v_scale(V, b->IORi)));
        if (v_dot(N, H) < 0.0f) H = v_neg(H);
    } else {
        H = v_normalize(v_add(L, V));
    }
    return H;
}

/* EvaluatePDF(lobe) BxDF.hlsli:287-299 */
static float bsdf_pdf(const bsdf_t *b, const surf_t *s, v3 L, v3 V, const float w[3], int lobe)
{
    v3 H = half_vector(b, s, L, V, w[LOBE_TRANSMISSION] > 0.0f);
    float lw = w[lobe];
    v3 N = s->Ns;
    if (lobe == LOBE_DIFFUSE) {
        if (v_dot(s->FrontNg, L) > 0.0f) { float nol = f_abs(v_dot(N, L)); return (nol * PT_INV_PI) * lw; }
        return 0.0f * lw;
    }
    if (lobe == LOBE_SPECULAR) {
        if (v_dot(s->FrontNg, L) > 0.0f) {
            v3 Vl = rotate_vector(s->basis, V);
            float vl[3] = { Vl.x, Vl.y, Vl.z };
            float noh = f_abs(v_dot(N, H));
            return oracle_vndf_pdf(vl, noh, b->Roughness) * lw;
        }
        return 0.0f * lw;
    }
    return f_abs(v_dot(N, L)) * lw;
}

/* Evaluate(lobe) BxDF.hlsli:301-315 */
static v3 bsdf_eval(const bsdf_t *b, const surf_t *s, v3 L, v3 V, const float w[3], int lobe)
{
    float wt = w[LOBE_TRANSMISSION];
    v3 H = half_vector(b, s, L, V, wt > 0.0f);
    v3 N = s->Ns;
    if (lobe == LOBE_TRANSMISSION) {
        float nol = f_abs(v_dot(N, L));
        return v_scale(v_scale(b->BaseColor, nol), wt);
    }
    float wr = 1.0f - wt;
    if (!(v_dot(s->FrontNg, L) > 0.0f)) return v_scale(V3(0.0f, 0.0f, 0.0f), wr);
    float nol = f_abs(v_dot(N, L)), nov = f_abs(v_dot(N, V)), voh = f_abs(v_dot(V, H));
    if (lobe == LOBE_DIFFUSE) {
        float dt = oracle_diffuse_term(b->Roughness, nol, nov, voh);
        return v_scale(v_scale(v_scale(b->Albedo, nol), dt), wr);
    }
    float noh = f_abs(v_dot(N, H));
    float D = oracle_distribution_term(b->Roughness, noh);
    float G = oracle_geometry_term_mod(b->Roughness, nol, nov);
    v3 F = fresnel_schlick(b->F0, voh);
    float k = nol * D * G;
    return v_scale(v_scale(F, k), wr);
}

void oracle_bsdf_step(const PtMaterial *m, int front, const float Ng_[3], const float V_[3],
                      const float rnd[4], OracleBsdfOut *out)
{
    bsdf_t b; surf_t s;
    v3 Ng = V3(Ng_[0], Ng_[1], Ng_[2]), V = V3(V_[0], V_[1], V_[2]);
    bsdf_init(&b, V3(m->BaseColor[0], m->BaseColor[1], m->BaseColor[2]), m->Metallic, m->Roughness, m->IOR, m->Transmission, front);
    surf_init(&s, front, Ng, front ? Ng : v_neg(Ng));
    lobe_weights(&b, &s, V, out->weights);
    v3 L = V3(0, 0, 0);
    out->valid = bsdf_sample(&b, &s, V, out->weights, rnd, &L, &out->lobe);
    out->L[0] = L.x; out->L[1] = L.y; out->L[2] = L.z;
    out->pdf = 0.0f; out->f[0] = out->f[1] = out->f[2] = 0.0f;
    if (out->valid) {
        out->pdf = bsdf_pdf(&b, &s, L, V, out->weights, out->lobe);
        v3 f = bsdf_eval(&b, &s, L, V, out->weights, out->lobe);
        out->f[0] = f.x; out->f[1] = f.y; out->f[2] = f.z;
    }
}

/* ------------------------------------------------------------------ the per-pixel loop */
typedef struct {
    float *events; uint32_t max_events, n_events;
} trace_t;

static void trace_event(trace_t *tr, uint32_t s, uint32_t bnc, const hit_t *h, int is_hit, v3 L, v3 T, uint32_t rng, int lobe, int flags)
{
    if (!tr || tr->n_events >= tr->max_events) return;
    float *e = tr->events + 16 * (size_t)tr->n_events++;
    e[0] = (float)s; e[1] = (float)bnc; e[2] = as_float(is_hit ? h->id : 0xFFFFFFFFu); e[3] = is_hit ? h->t : INFINITY;
    e[4] = is_hit ? h->P.x : 0; e[5] = is_hit ? h->P.y : 0; e[6] = is_hit ? h->P.z : 0;
    e[7] = L.x; e[8] = L.y; e[9] = L.z; e[10] = T.x; e[11] = T.y; e[12] = T.z;
    e[13] = as_float(rng); e[14] = (float)lobe; e[15] = (float)flags;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Row N1: textured spheres.  EvaluateMaterial (Shaders/ShadingHelpers.hlsli:161-235) with Sample (:53-59), EvaluateBaseColor
 * (:61-72), EvaluateTransmission (:74-85), PerturbNormal (:87-103), Math::CalculateTBN (Math.hlsli:17-21),
 * HitInfo::GetFrontTangent (HitInfo.hlsli:91-94).  The mesh-derived inputs are restated for analytic spheres (DESIGN.md
 * specs S6-S8): GeoSphere texture coordinates from the object-space normal, tangent along increasing u, level-0 bilinear
 * wrap sampling, and a fixed-polynomial atan2.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct tex_ctx_s {
    const OracleTextures *t;
    float unorm[256], srgb[256]; /* 8-bit code -> linear value */
} tex_ctx;

static void tex_ctx_init(tex_ctx *c, const OracleTextures *t)
{
    c->t = t;
    for (int v = 0; v < 256; v++) {
        c->unorm[v] = (float)v * (1.0f / 255.0f);
        c->srgb[v] = oracle_from_srgb(c->unorm[v]);
    }
}

static size_t sizeof_tex_ctx(void) { return sizeof(tex_ctx); }
static void tex_ctx_init_opaque(struct tex_ctx_s *c, const OracleTextures *t) { tex_ctx_init(c, t); }

float oracle_atan2(float y, float x)
{
    float ax = f_abs(x), ay = f_abs(y);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;
    float a = mn / mx, s = a * a;
    float r = a * FMA(s, FMA(s, FMA(s, FMA(s, FMA(s, -0.01172120f, 0.05265332f), -0.11643287f), 0.19354346f), -0.33262347f), 0.99997726f);
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return y < 0.0f ? -r : r;
}

void oracle_sphere_uv(const float n[3], float uv[2])
{
    float lon = oracle_atan2(n[0], -n[2]);
    float lat = oracle_atan2(sqrtf(f_max(FMA(-n[1], n[1], 1.0f), 0.0f)), n[1]);
    uv[0] = 1.0f - FMA(lon, 0.15915494309189533577f, 0.5f);
    uv[1] = lat * 0.31830988618379067154f;
}

static v3 v_cross(v3 a, v3 b)
{
    return V3(FMA(a.y, b.z, -(a.z * b.y)), FMA(a.z, b.x, -(a.x * b.z)), FMA(a.x, b.y, -(a.y * b.x)));
}

static v3 quat_rotate(float qx, float qy, float qz, float qw, v3 v)
{
    v3 u = V3(qx, qy, qz);
    v3 t = v_scale(v_cross(u, v), 2.0f);
    return v_add(v_mad(qw, t, v), v_cross(u, t));
}

void oracle_quat_rotate(const float q[4], const float v[3], float out[3])
{
    v3 r = quat_rotate(q[0], q[1], q[2], q[3], V3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static v3 sphere_tangent(v3 n)
{
    float l2 = FMA(n.z, n.z, n.x * n.x);
    if (!(l2 > 0.0f)) return V3(0, 0, 0);
    float inv = 1.0f / sqrtf(l2);
    return V3(n.z * inv, 0.0f, -n.x * inv);
}

void oracle_sphere_tangent(const float n[3], float t[3])
{
    v3 r = sphere_tangent(V3(n[0], n[1], n[2]));
    t[0] = r.x; t[1] = r.y; t[2] = r.z;
}

static uint32_t wrap_index(int i, uint32_t n) { int m = i % (int)n; return (uint32_t)(m < 0 ? m + (int)n : m); }
static float lerp1(float a, float b, float t) { return FMA(t, b - a, a); }

static void fetch_texel(const tex_ctx *c, const PtTexture *tx, uint32_t x, uint32_t y, float out[4])
{
    if (tx->Format == PT_TEXTURE_RGBA32_FLOAT) { /* linear HDR texels (environment maps) */
        const float *f = (const float *)tx->Pixels + 4u * ((size_t)y * tx->Width + x);
        out[0] = f[0]; out[1] = f[1]; out[2] = f[2]; out[3] = f[3];
        return;
    }
    const uint8_t *p = (const uint8_t *)tx->Pixels + 4u * ((size_t)y * tx->Width + x);
    const float *lut = tx->Format == PT_TEXTURE_RGBA8_UNORM_SRGB ? c->srgb : c->unorm;
    out[0] = lut[p[0]]; out[1] = lut[p[1]]; out[2] = lut[p[2]]; out[3] = c->unorm[p[3]];
}

/* SampleLevel(g_anisotropicSampler, uv, 0): level-0 bilinear, wrap */
static void sample_bilinear(const tex_ctx *c, uint32_t index, const float uv[2], float out[4])
{
    const PtTexture *tx = &c->t->textures[index];
    float u = uv[0], v = uv[1];
    if (!(f_abs(u) < 65536.0f)) u = 0.0f;
    if (!(f_abs(v) < 65536.0f)) v = 0.0f;
    float x = FMA(u, (float)tx->Width, -0.5f), y = FMA(v, (float)tx->Height, -0.5f);
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    uint32_t x0 = wrap_index((int)xf, tx->Width), x1 = wrap_index((int)xf + 1, tx->Width);
    uint32_t y0 = wrap_index((int)yf, tx->Height), y1 = wrap_index((int)yf + 1, tx->Height);
    float c00[4], c10[4], c01[4], c11[4];
    fetch_texel(c, tx, x0, y0, c00); fetch_texel(c, tx, x1, y0, c10);
    fetch_texel(c, tx, x0, y1, c01); fetch_texel(c, tx, x1, y1, c11);
    for (int k = 0; k < 4; k++) out[k] = lerp1(lerp1(c00[k], c10[k], fx), lerp1(c01[k], c11[k], fx), fy);
}

void oracle_sample_texture(const OracleTextures *t, uint32_t index, const float uv[2], float out[4])
{
    tex_ctx c;
    tex_ctx_init(&c, t);
    sample_bilinear(&c, index, uv, out);
}

/* GetEnvironmentLightColor's texture branch (ShadingHelpers.hlsli:13-24) for a lat-long map: the direction is rotated by the
 * upper 3x3 of EnvironmentLightTransform (Geometry::RotateVector = mul(M, v): component i = dot(row i, v)), normalised, mapped
 * by Math::ToLatLongCoordinate (Math.hlsli:29-33: u = (1 + atan2(x, z) / pi) / 2, v = acos(y) / pi) and sampled at level 0. */
void oracle_latlong_uv(const float d[3], float uv[2])
{
    uv[0] = FMA(oracle_atan2(d[0], d[2]), 0.15915494309189533577f, 0.5f);
    uv[1] = oracle_atan2(sqrtf(f_max(FMA(-d[1], d[1], 1.0f), 0.0f)), d[1]) * 0.31830988618379067154f;
}

/* TextureCube::SampleLevel(sampler, d, 0) (ShadingHelpers.hlsli:17-21), spec S9: face = axis of largest magnitude (ties z over
 * y over x), D3D face order +X -X +Y -Y +Z -Z and face coordinates, level-0 bilinear inside the face, clamp addressing. */
uint32_t oracle_cube_face_uv(const float d[3], float uv[2])
{
    float ax = f_abs(d[0]), ay = f_abs(d[1]), az = f_abs(d[2]);
    uint32_t face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) { face = d[2] < 0.0f ? 5u : 4u; sc = d[2] < 0.0f ? -d[0] : d[0]; tc = -d[1]; ma = az; }
    else if (ay >= ax)        { face = d[1] < 0.0f ? 3u : 2u; sc = d[0]; tc = d[1] < 0.0f ? -d[2] : d[2]; ma = ay; }
    else                      { face = d[0] < 0.0f ? 1u : 0u; sc = d[0] < 0.0f ? d[2] : -d[2]; tc = -d[1]; ma = ax; }
    uv[0] = FMA(sc / ma, 0.5f, 0.5f);
    uv[1] = FMA(tc / ma, 0.5f, 0.5f);
    return face;
}

static uint32_t clamp_index(int i, uint32_t n) { return i < 0 ? 0u : ((uint32_t)i >= n ? n - 1u : (uint32_t)i); }

static void sample_bilinear_clamp(const tex_ctx *c, uint32_t index, const float uv[2], float out[4])
{
    const PtTexture *tx = &c->t->textures[index];
    float u = uv[0], v = uv[1];
    if (!(f_abs(u) < 65536.0f)) u = 0.0f;
    if (!(f_abs(v) < 65536.0f)) v = 0.0f;
    float x = FMA(u, (float)tx->Width, -0.5f), y = FMA(v, (float)tx->Height, -0.5f);
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    uint32_t x0 = clamp_index((int)xf, tx->Width), x1 = clamp_index((int)xf + 1, tx->Width);
    uint32_t y0 = clamp_index((int)yf, tx->Height), y1 = clamp_index((int)yf + 1, tx->Height);
    float c00[4], c10[4], c01[4], c11[4];
    fetch_texel(c, tx, x0, y0, c00); fetch_texel(c, tx, x1, y0, c10);
    fetch_texel(c, tx, x0, y1, c01); fetch_texel(c, tx, x1, y1, c11);
    for (int k = 0; k < 4; k++) out[k] = lerp1(lerp1(c00[k], c10[k], fx), lerp1(c01[k], c11[k], fx), fy);
}

static v3 environment_texture(const tex_ctx *tc, const PtSceneData *sd, v3 d)
{
    const float *m = sd->EnvironmentLightTransform;
    v3 r = v_normalize(V3(v_dot(V3(m[0], m[1], m[2]), d), v_dot(V3(m[4], m[5], m[6]), d), v_dot(V3(m[8], m[9], m[10]), d)));
    float dir[3] = { r.x, r.y, r.z }, uv[2], s[4];
    if (sd->IsEnvironmentLightTextureCubeMap) {
        uint32_t face = oracle_cube_face_uv(dir, uv);
        sample_bilinear_clamp(tc, sd->EnvironmentLightTextureDescriptor + face, uv, s);
    } else {
        oracle_latlong_uv(dir, uv);
        sample_bilinear(tc, sd->EnvironmentLightTextureDescriptor, uv, s);
    }
    return V3(s[0], s[1], s[2]);
}

/* PerturbNormal: Geometry::UnpackLocalNormal (MathLib, recollection) + CalculateTBN + RotateVectorInverse */
static v3 perturb_normal(v3 N, v3 T, float sx, float sy)
{
    const float k = 255.0f / 127.0f;
    float x = FMA(sx, k, -1.0f), y = FMA(sy, k, -1.0f);
    float z = f_sqrt01(1.0f - FMA(y, y, x * x));
    v3 Tn = v_normalize(v_sub(T, v_scale(N, v_dot(N, T))));
    v3 B = v_cross(N, Tn);
    return v_normalize(v_mad(x, Tn, v_mad(y, B, v_scale(N, z))));
}

void oracle_perturb_normal(const float N[3], const float T[3], float sx, float sy, float out[3])
{
    v3 r = perturb_normal(V3(N[0], N[1], N[2]), V3(T[0], T[1], T[2]), sx, sy);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* texture coordinates of the point of sphere `id` whose outward world-space normal is N (spec S6); also returns the object's
 * rotation q and the mesh-space normal nn */
static void hit_uv(const tex_ctx *tc, uint32_t id, v3 N, float q[4], float nn[3], float uv[2])
{
    q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 1;
    if (tc->t->rotations) memcpy(q, tc->t->rotations + 4u * (size_t)id, 4 * sizeof(float));
    v3 n_obj = quat_rotate(-q[0], -q[1], -q[2], q[3], N); /* world -> object */
    /* ObjectToWorld = diag(1, 1, -1) * pose (Scene.ixx:197-199): the mesh-space normal is the z mirror of the object-space one (the
     * rotations handed over are the poses conjugated with that mirror).  Settled against Screenshots/Raytracing-Spheres.png with the
     * reference's own Earth map: without the mirror the continents come out mirrored. */
    nn[0] = n_obj.x; nn[1] = n_obj.y; nn[2] = -n_obj.z;
    oracle_sphere_uv(nn, uv);
}

/* IsOpaque (ShadingHelpers.hlsli:105-115) for the crossing at parameter t of sphere `id` (AlphaMode != Opaque): alpha =
 * BaseColor.a, times the base-colour map's alpha at the crossing when EvaluateBaseColor samples (:61-72: any component of the
 * float4 BaseColor > 0 and a map); accepted iff alpha >= AlphaCutoff (NaN: rejected). */
static int crossing_is_opaque(const alpha_ctx *a, const PtSphere *sp, uint32_t id, v3 o, v3 d, float t)
{
    const PtMaterial *m = &a->mat[id];
    float alpha = m->BaseColor[3];
    const tex_ctx *tc = a->tc;
    if (tc && tc->t->object_textures) {
        const uint32_t desc = tc->t->object_textures[id].Maps[PT_TEXTURE_MAP_BASE_COLOR].Descriptor;
        if (desc != ~0u && (m->BaseColor[0] > 0.0f || m->BaseColor[1] > 0.0f || m->BaseColor[2] > 0.0f || m->BaseColor[3] > 0.0f)) {
            v3 C = V3(sp->cx, sp->cy, sp->cz);
            v3 N = v_normalize(v_sub(v_mad(t, d, o), C)); /* the hit frame's normal (cast_ray) */
            float q[4], nn[3], uv[2], s[4];
            hit_uv(tc, id, N, q, nn, uv);
            sample_bilinear(tc, desc, uv, s);
            alpha = alpha * s[3];
        }
    }
    return alpha >= m->AlphaCutoff;
}

typedef struct {
    v3 base, emissive_color;
    float emissive_strength, metallic, roughness, ior, transmission;
    v3 shadingN;
} material_eval;

/* The material of a hit: constant material, modulated by the object's texture maps (if any) at the hit's texture
 * coordinates; shading normal perturbed by its normal map. */
static material_eval evaluate_material(const tex_ctx *tc, const PtMaterial *m, const hit_t *h)
{
    material_eval e;
    e.base = V3(m->BaseColor[0], m->BaseColor[1], m->BaseColor[2]);
    e.emissive_color = V3(m->EmissiveColor[0], m->EmissiveColor[1], m->EmissiveColor[2]);
    e.emissive_strength = m->EmissiveStrength;
    e.metallic = m->Metallic; e.roughness = m->Roughness; e.ior = m->IOR; e.transmission = m->Transmission;
    e.shadingN = h->shadingN;
    if (!tc || !tc->t->object_textures) return e;
    const PtTextureMapInfo *maps = tc->t->object_textures[h->id].Maps;
    int any = 0;
    for (int k = 0; k < PT_TEXTURE_MAP_COUNT; k++) any |= maps[k].Descriptor != ~0u;
    if (!any) return e;

    float q[4], nn[3], uv[2], s[4];
    hit_uv(tc, h->id, h->N, q, nn, uv);
    v3 t_mesh = sphere_tangent(V3(nn[0], nn[1], nn[2]));
    v3 T = quat_rotate(q[0], q[1], q[2], q[3], V3(t_mesh.x, t_mesh.y, -t_mesh.z));
    if (!h->front) T = v_neg(T); /* GetFrontTangent */

    if ((e.base.x > 0.0f || e.base.y > 0.0f || e.base.z > 0.0f) && maps[PT_TEXTURE_MAP_BASE_COLOR].Descriptor != ~0u) {
        sample_bilinear(tc, maps[PT_TEXTURE_MAP_BASE_COLOR].Descriptor, uv, s);
        e.base = V3(e.base.x * s[0], e.base.y * s[1], e.base.z * s[2]);
    }
    v3 emission = v_scale(e.emissive_color, e.emissive_strength);
    if ((emission.x > 0.0f || emission.y > 0.0f || emission.z > 0.0f) && maps[PT_TEXTURE_MAP_EMISSIVE_COLOR].Descriptor != ~0u) {
        sample_bilinear(tc, maps[PT_TEXTURE_MAP_EMISSIVE_COLOR].Descriptor, uv, s);
        e.emissive_color = V3(e.emissive_color.x * s[0], e.emissive_color.y * s[1], e.emissive_color.z * s[2]);
    }
    if (maps[PT_TEXTURE_MAP_METALLIC_ROUGHNESS].Descriptor != ~0u) {
        if (m->Metallic > 0.0f || m->Roughness > 0.0f) {
            sample_bilinear(tc, maps[PT_TEXTURE_MAP_METALLIC_ROUGHNESS].Descriptor, uv, s);
            e.metallic = m->Metallic * s[2];
            e.roughness = m->Roughness * s[1];
        }
    } else {
        if (m->Metallic > 0.0f && maps[PT_TEXTURE_MAP_METALLIC].Descriptor != ~0u) {
            sample_bilinear(tc, maps[PT_TEXTURE_MAP_METALLIC].Descriptor, uv, s);
            e.metallic = m->Metallic * s[0];
        }
        if (m->Roughness > 0.0f && maps[PT_TEXTURE_MAP_ROUGHNESS].Descriptor != ~0u) {
            sample_bilinear(tc, maps[PT_TEXTURE_MAP_ROUGHNESS].Descriptor, uv, s);
            e.roughness = m->Roughness * s[0];
        }
    }
    if (e.metallic < 1.0f && m->Transmission > 0.0f && maps[PT_TEXTURE_MAP_TRANSMISSION].Descriptor != ~0u) {
        sample_bilinear(tc, maps[PT_TEXTURE_MAP_TRANSMISSION].Descriptor, uv, s);
        e.transmission = m->Transmission * s[0];
    }
    if ((T.x != 0.0f || T.y != 0.0f || T.z != 0.0f) && maps[PT_TEXTURE_MAP_NORMAL].Descriptor != ~0u) {
        sample_bilinear(tc, maps[PT_TEXTURE_MAP_NORMAL].Descriptor, uv, s);
        e.shadingN = perturb_normal(h->shadingN, T, s[0], s[1]);
    }
    return e;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Row N4: sphere-light direct illumination of the primary surface -- a functional stand-in for the RTXDI passes whose
 * result Raytracing.hlsl:150-163 reads (DI = directDiffuse + directSpecular, isDIValid = any(DI > 0)), :302 (the first-bounce
 * hit of a pixel with valid DI contributes no emission) and :381 (radiance += DI).  One emitter chosen uniformly
 * (LightPreparation.ixx:52-70 enumerates them), one direction uniform in the cone it subtends, own RNG stream.
 * ---------------------------------------------------------------------------------------------------------------- */
#define DI_NEGLIGIBLE 1e-7f
#define DI_RNG_SALT 0x44495F31u

typedef struct { const uint32_t *ids; uint32_t n; } light_list;

int oracle_sample_sphere_cone(const float P[3], const float C[3], float r, float u1, float u2, float L[3], float *inv_pdf)
{
    v3 w = v_sub(V3(C[0], C[1], C[2]), V3(P[0], P[1], P[2]));
    float d2 = v_dot(w, w), r2 = r * r;
    L[0] = 0; L[1] = 0; L[2] = 1; *inv_pdf = 0.0f;
    if (!(d2 > r2)) return 0;
    v3 wn = v_scale(w, 1.0f / sqrtf(d2));
    float sin2 = r2 / d2;
    float cos_max = f_sqrt01(1.0f - sin2);
    float omc = sin2 / (1.0f + cos_max);
    float k = u1 * omc;
    float cos_t = 1.0f - k;
    float sin_t = f_sqrt01(k * (1.0f + cos_t));
    float sp, cp;
    oracle_sincos_2pi(u2, &sp, &cp);
    basis3 b = get_basis(wn);
    v3 l = rotate_vector_inverse(b, V3(sin_t * cp, sin_t * sp, cos_t));
    L[0] = l.x; L[1] = l.y; L[2] = l.z;
    *inv_pdf = 6.28318530717958647692f * omc;
    return 1;
}

/* Raytracing.hlsl:103-415 (DEFAULT permutation) + GBufferGeneration.hlsl:117-232 primary hit.
 * Writes rgba; returns rays cast. */
static uint64_t render_pixel(const PtSphere *sph, const PtMaterial *mat, uint32_t n,
                             const PtSceneData *sd, const PtCamera *cam, const PtGraphicsSettings *gs,
                             uint32_t px, uint32_t py, float rgba[4], uint64_t *paths_out, trace_t *tr, const tex_ctx *tc,
                             const light_list *lights, const void *accel, int has_alpha)
{
    (void)paths_out;
    uint64_t rays = 0;
    uint32_t rng = oracle_rng_init(px, py, gs->FrameIndex); /* :108 */
    v3 o, d; float tmin, tmax;
    primary_ray(cam, px, py, gs->RenderSize[0], gs->RenderSize[1], &o, &d, &tmin, &tmax); /* :138 */

    /* alpha-tested hits (spec S10) only exist when some object is not Opaque */
    const alpha_ctx actx = { mat, tc };
    const alpha_ctx *alpha = has_alpha ? &actx : NULL;

    /* primary-hit pass (GBufferGeneration.hlsl:128-230) */
    hit_t primary;
    cast_ray(accel, sph, n, o, d, tmin, tmax, &primary, alpha);
    rays++;
    const int di_on = gs->IsDIEnabled && lights && lights->n > 0;
    if (!primary.hit) { /* miss: Radiance = env (GBufferGeneration.hlsl:223-227); bounce loop returns without writing (:249-252) */
        v3 c = environment_color(tc, sd, d);
        rgba[0] = c.x; rgba[1] = c.y; rgba[2] = c.z; rgba[3] = 1.0f;
        trace_event(tr, 0, 0, &primary, 0, V3(0, 0, 0), V3(1, 1, 1), rng, -1, 1);
        return rays;
    }
    /* GBufferGeneration.hlsl:152-166: EvaluateMaterial at the primary hit (texture maps, normal map), then Initialize */
    const material_eval pe = evaluate_material(tc, &mat[primary.id], &primary);
    primary.shadingN = pe.shadingN;
    v3 primary_radiance = v_scale(pe.emissive_color, pe.emissive_strength);
    bsdf_t primary_bsdf;
    bsdf_init(&primary_bsdf, pe.base, pe.metallic, pe.roughness, pe.ior,
              pe.metallic < 1.0f ? pe.transmission : 0.0f, primary.front); /* :143-150 */

    /* direct illumination of the primary surface (row N4) */
    v3 DI = V3(0, 0, 0);
    int di_valid = 0;
    if (di_on) {
        uint32_t lrng = oracle_rng_init(px, py, gs->FrameIndex ^ DI_RNG_SALT);
        float u0 = oracle_rng_float(&lrng), u1 = oracle_rng_float(&lrng), u2 = oracle_rng_float(&lrng);
        uint32_t j = (uint32_t)(u0 * (float)lights->n);
        if (j >= lights->n) j = lights->n - 1u;
        const uint32_t light = lights->ids[j];
        const PtSphere *ls = &sph[light];
        float Pp[3] = { primary.P.x, primary.P.y, primary.P.z }, Cc[3] = { ls->cx, ls->cy, ls->cz }, Ll[3], inv_pdf;
        int ok = oracle_sample_sphere_cone(Pp, Cc, ls->r, u1, u2, Ll, &inv_pdf);
        v3 L = V3(Ll[0], Ll[1], Ll[2]);
        surf_t sv;
        surf_init(&sv, primary.front, primary.N, primary.shadingN);
        if (light != primary.id && ok && v_dot(sv.FrontNg, L) > 0.0f) {
            v3 V = v_neg(d);
            float w[3];
            lobe_weights(&primary_bsdf, &sv, V, w);
            v3 f = v_add(bsdf_eval(&primary_bsdf, &sv, L, V, w, 0), bsdf_eval(&primary_bsdf, &sv, L, V, w, 1));
            /* no shadow ray for a contribution that cannot matter (upper bound with the emitter's untextured radiance <= 1e-7) */
            const PtMaterial *lmc = &mat[light];
            const float kk = inv_pdf * (float)lights->n;
            const float bound = f_max(f.x * lmc->EmissiveColor[0], f_max(f.y * lmc->EmissiveColor[1], f.z * lmc->EmissiveColor[2])) * (lmc->EmissiveStrength * kk);
            hit_t sh;
            sh.hit = 0;
            if (bound > DI_NEGLIGIBLE) {
                cast_ray(accel, sph, n, spawn_origin(primary.P, primary.N, primary.offset, L), L, 0.0f, INFINITY, &sh, alpha);
                rays++;
            }
            if (sh.hit && sh.id == light) {
                /* the emitter's radiance at the point the shadow ray reaches: Material::GetEmission after EvaluateMaterial (an
                 * emissive map modulates the constant; LightPreparation.hlsl:84-88 likewise reads the map for its triangles) */
                const material_eval lme = evaluate_material(tc, &mat[light], &sh);
                v3 le = v_scale(lme.emissive_color, lme.emissive_strength);
                DI = v_scale(v_mul(le, f), kk);
            }
        }
        /* NaN / inf / negative estimates count as no light.  The reference gates :302 on any(DI > 0); with this one-sample
         * estimator DI = 0 is an ordinary sample value, so the gate is "the pixel has a primary surface" (keeps it unbiased) */
        if (!(DI.x > 0.0f || DI.y > 0.0f || DI.z > 0.0f) || !f_finite(DI.x) || !f_finite(DI.y) || !f_finite(DI.z)) DI = V3(0, 0, 0);
        di_valid = 1;
    }

    v3 radiance = V3(0, 0, 0);
    const uint32_t spp = gs->SamplesPerPixel;
    for (uint32_t s = 0; s < spp; s++) { /* :191 */
        v3 ro = o, rd = d;
        int is_hit = 1;
        hit_t hit = primary;
        v3 emission = primary_radiance;
        bsdf_t bsdf = primary_bsdf;
        int lobe = -1;
        v3 L = V3(0, 0, 0), T = V3(1, 1, 1);
        v3 sample_radiance = V3(0, 0, 0);
        int via_t = 0;
        for (uint32_t bnc = 0; bnc <= gs->Bounces; bnc++) { /* :213 */
            if (bnc) { /* :219-234 */
                ro = spawn_origin(hit.P, hit.N, hit.offset, L);
                rd = L;
                cast_ray(accel, sph, n, ro, rd, 0.0f, INFINITY, &hit, alpha);
                is_hit = hit.hit;
                rays++;
            }
            if (!is_hit) { /* :242-259 (bnc > 0 here) */
                v3 env = environment_color(tc, sd, rd);
                sample_radiance = v_add(sample_radiance, v_mul(T, env));
                trace_event(tr, s, bnc, &hit, 0, L, T, rng, lobe, 2);
                break;
            }
            if (bnc) { /* :293-305 */
                const material_eval e = evaluate_material(tc, &mat[hit.id], &hit); /* :293-301 */
                hit.shadingN = e.shadingN;
                emission = v_scale(e.emissive_color, e.emissive_strength);
                /* :302 -- DI covers what the primary surface's reflective lobes receive; a sample that left it through the
                 * transmission lobe keeps the emission it finds */
                if (di_valid && bnc == 1 && !via_t) emission = V3(0, 0, 0);
                bsdf_init(&bsdf, e.base, e.metallic, e.roughness, e.ior, e.transmission, hit.front);
            }
            sample_radiance = v_add(sample_radiance, v_mul(T, emission)); /* :320 */

            surf_t sv;
            surf_init(&sv, hit.front, hit.N, hit.shadingN); /* :323-324 */
            v3 V = v_neg(rd);
            float w[3];
            lobe_weights(&bsdf, &sv, V, w); /* :329 */
            float rnd[4];
            rnd[0] = oracle_rng_float(&rng); rnd[1] = oracle_rng_float(&rng);
            rnd[2] = oracle_rng_float(&rng); rnd[3] = oracle_rng_float(&rng); /* GetFloat4 :330 */
            if (!bsdf_sample(&bsdf, &sv, V, w, rnd, &L, &lobe)) { trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 3); break; }
            float pdf = bsdf_pdf(&bsdf, &sv, L, V, w, lobe); /* :335 */
            if (pdf == 0.0f) { trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 4); break; }
            v3 f = bsdf_eval(&bsdf, &sv, L, V, w, lobe); /* :341 */
            if (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f) { trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 5); break; }
            { float inv_pdf = 1.0f / pdf; T = v_mul(T, v_scale(f, inv_pdf)); } /* :346 */
            if (bnc == 0) via_t = lobe == 2;
            if (gs->IsRussianRouletteEnabled && bnc > 3) { /* :348-356 */
                float p = f_max(T.x, f_max(T.y, T.z));
                if (oracle_rng_float(&rng) >= p) { trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 6); break; }
                T = v_scale(T, 1.0f / p);
            }
            if (luminance(T) <= gs->ThroughputThreshold) { trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 7); break; } /* :361 */
            trace_event(tr, s, bnc, &hit, 1, L, T, rng, lobe, 0);
        }
        radiance = v_add(radiance, sample_radiance); /* :373 */
    }
    if (f_finite(radiance.x) && f_finite(radiance.y) && f_finite(radiance.z)) { /* :378 */
        radiance = v_scale(radiance, 1.0f / (float)spp);
    } else {
        radiance = V3(0, 0, 0);
    }
    if (di_on) radiance = v_add(radiance, DI); /* :381 */
    rgba[0] = radiance.x; rgba[1] = radiance.y; rgba[2] = radiance.z; rgba[3] = 1.0f;
    return rays;
}

typedef struct {
    const PtSphere *sph; const PtMaterial *mat; uint32_t n;
    const PtSceneData *sd; const PtCamera *cam; const PtGraphicsSettings *gs;
    PtRect rect; uint32_t row_step; float *out;
    const void *tex; /* tex_ctx */
    const void *lights; /* light_list */
    const void *accel;  /* obvh or NULL (brute force) */
    int has_alpha;      /* some object's AlphaMode is not Opaque */
    int tid, nthreads;
    uint64_t rays, paths;
} job_t;

/* ------------------------------------------------------------------------------------------------------------------
 * Row N3: display transform + progressive accumulation.
 * App::Impl::ToneMap (Source/App.cpp:1731-1757) -> DirectXTK ToneMapPostProcess (un-vendored; operator / transfer pairs
 * created at Source/App.cpp:760-769).  Shader arithmetic restated from the published ToneMap.fx (recollection):
 *   SDR : c = hdr * linearExposure; Saturate | Reinhard c/(1+c) | ACESFilmic saturate(c(2.51c+.03)/(c(2.43c+.59)+.14));
 *         SRGB: pow(|c|, 1/2.2)  -> R8G8B8A8_UNORM
 *   HDR10: c = rotation * hdr; LinearToST2084(c * paperWhite / 10000)  -> R10G10B10A2_UNORM
 * ---------------------------------------------------------------------------------------------------------------- */
static float pow_pos(float x, float y) { return x > 0.0f ? oracle_pow(x, y) : 0.0f; }

static float tone_operator(float x, uint32_t op)
{
    switch (op) {
    case 1: return f_sat(x);
    case 2: return x / (1.0f + x);
    case 3: return f_sat((x * FMA(2.51f, x, 0.03f)) / FMA(x, FMA(2.43f, x, 0.59f), 0.14f));
    default: return x;
    }
}

static float srgb_est(float c) { return pow_pos(f_sat(c), 1.0f / 2.2f); }

static float st2084(float y)
{
    float ym = pow_pos(f_min(f_abs(y), 1.0e30f), 0.1593017578f);
    return pow_pos(FMA(18.8515625f, ym, 0.8359375f) / FMA(18.6875f, ym, 1.0f), 78.84375f);
}

static uint32_t to_unorm(float v, float scale) { return (uint32_t)FMA(f_sat(v), scale, 0.5f); }

static const float k_rotation[3][9] = {
    { 0.6274040f, 0.3292820f, 0.0433136f, 0.0690970f, 0.9195400f, 0.0113612f, 0.0163916f, 0.0880132f, 0.8955950f },     /* 709 -> 2020 */
    { 0.753845f, 0.198593f, 0.047562f, 0.0457456f, 0.941777f, 0.0124772f, -0.00121055f, 0.0176041f, 0.983607f },        /* P3-D65 -> 2020 */
    { 0.822461969f, 0.1775380f, 0.0f, 0.033194199f, 0.966805801f, 0.0f, 0.017082631f, 0.0723974f, 0.910519969f },       /* 709 -> P3-D65 */
};

uint32_t oracle_tonemap_pixel(const float hdr[3], const PtToneMapParams *p)
{
    if (p->TransferFunction == 2) {
        const float *m = k_rotation[p->ColorRotation < 3 ? p->ColorRotation : 0];
        v3 c = V3(hdr[0], hdr[1], hdr[2]);
        float k = p->PaperWhiteNits * (1.0f / 10000.0f);
        float r = st2084(v_dot(V3(m[0], m[1], m[2]), c) * k);
        float g = st2084(v_dot(V3(m[3], m[4], m[5]), c) * k);
        float b = st2084(v_dot(V3(m[6], m[7], m[8]), c) * k);
        return to_unorm(r, 1023.0f) | (to_unorm(g, 1023.0f) << 10) | (to_unorm(b, 1023.0f) << 20) | (3u << 30);
    }
    float c[3];
    for (int i = 0; i < 3; i++) {
        c[i] = tone_operator(hdr[i] * p->LinearExposure, p->Operator);
        if (p->TransferFunction == 1) c[i] = srgb_est(c[i]);
    }
    return to_unorm(c[0], 255.0f) | (to_unorm(c[1], 255.0f) << 8) | (to_unorm(c[2], 255.0f) << 16) | (255u << 24);
}

void oracle_tonemap(const float *hdr_rgba, uint32_t n_pixels, const PtToneMapParams *p, uint32_t *out)
{
    for (uint32_t i = 0; i < n_pixels; i++) out[i] = oracle_tonemap_pixel(hdr_rgba + 4u * i, p);
}

/* running mean over frames: n = frames already accumulated */
void oracle_accumulate(float *accum_rgba, const float *radiance_rgba, uint32_t n_pixels, uint32_t frames_accumulated)
{
    float inv = 1.0f / (float)(frames_accumulated + 1u);
    for (uint32_t i = 0; i < 4u * n_pixels; i++)
        accum_rgba[i] = frames_accumulated == 0 ? radiance_rgba[i] : FMA(radiance_rgba[i] - accum_rgba[i], inv, accum_rgba[i]);
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    uint32_t k = 0;
    for (uint32_t ry = 0; ry < j->rect.h; ry += j->row_step, k++) {
        if ((int)(k % (uint32_t)j->nthreads) != j->tid) continue;
        for (uint32_t rx = 0; rx < j->rect.w; rx++) {
            float *px = j->out + 4 * ((size_t)ry * j->rect.w + rx);
            j->rays += render_pixel(j->sph, j->mat, j->n, j->sd, j->cam, j->gs, j->rect.x + rx, j->rect.y + ry, px, &j->paths, NULL, (const tex_ctx *)j->tex, (const light_list *)j->lights, j->accel, j->has_alpha);
            j->paths += j->gs->SamplesPerPixel; /* nominal (pixel, sample) pairs */
        }
    }
    return NULL;
}

static int validate(const PtSceneData *sd, const PtGraphicsSettings *gs, uint32_t n)
{
    (void)n; /* n == 0 is a legal scene: a TLAS without instances -- every ray misses, every pixel is the environment */
    if (gs->RenderSize[0] == 0 || gs->RenderSize[1] == 0 || gs->SamplesPerPixel == 0) return 2;
    if (gs->RenderSize[0] > 65535u || gs->RenderSize[1] > 65535u) return 2; /* (px<<16)|py seed */
    if (gs->Denoiser) return 3;
    return 0;
}

static int validate_textures(const OracleTextures *t, uint32_t n)
{
    if (!t || t->n_textures == 0) return 0;
    if (!t->textures) return 6;
    for (uint32_t i = 0; i < t->n_textures; i++)
        if (!t->textures[i].Pixels || !t->textures[i].Width || !t->textures[i].Height || t->textures[i].Format > PT_TEXTURE_RGBA32_FLOAT) return 6;
    if (!t->object_textures) return 0; /* an environment map alone */
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < PT_TEXTURE_MAP_COUNT; k++) {
            const PtTextureMapInfo *mi = &t->object_textures[i].Maps[k];
            if (mi->Descriptor != ~0u && (mi->Descriptor >= t->n_textures || mi->TextureCoordinateIndex != 0)) return 6;
        }
    return 0;
}

int oracle_render(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                  const PtSceneData *scene, const PtCamera *camera,
                  const PtGraphicsSettings *gs, const PtRect *rect, uint32_t row_step,
                  float *out_rgba, OracleStats *stats, int threads)
{
    return oracle_render_textured(spheres, materials, n, scene, camera, gs, rect, row_step, out_rgba, stats, threads, NULL);
}

int oracle_render_textured(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                           const PtSceneData *scene, const PtCamera *camera,
                           const PtGraphicsSettings *gs, const PtRect *rect, uint32_t row_step,
                           float *out_rgba, OracleStats *stats, int threads, const OracleTextures *textures)
{
    int err = validate(scene, gs, n);
    if (err) return err;
    if ((err = validate_textures(textures, n)) != 0) return err;
    if (rect->x + rect->w > gs->RenderSize[0] || rect->y + rect->h > gs->RenderSize[1]) return 5;
    if (scene->EnvironmentLightTextureDescriptor != 0xFFFFFFFFu) { /* the descriptor indexes the texture table; a cube map = 6 square faces */
        const uint32_t e = scene->EnvironmentLightTextureDescriptor, nf = scene->IsEnvironmentLightTextureCubeMap ? 6u : 1u;
        if (!textures || (uint64_t)e + nf > textures->n_textures) return 4;
        for (uint32_t f = 0; f < nf && nf == 6u; f++)
            if (textures->textures[e + f].Width != textures->textures[e].Width || textures->textures[e + f].Height != textures->textures[e].Width) return 4;
    }
    tex_ctx tc;
    const int textured = textures && textures->n_textures > 0;
    if (textured) tex_ctx_init(&tc, textures);
    /* ORACLE_NO_BVH=1 forces the brute-force definition (the tests compare whole frames both ways).  The structure of the
     * last scene is kept between calls (keyed by a hash of the sphere data), so that timing consecutive frames of one scene
     * -- bench.py's cpu_baseline -- measures rendering, as the GPU figure does, not rebuilding.  Not re-entrant: one
     * oracle_render at a time per process, which is how tests/ and bench.py use it. */
    static obvh *cached = NULL;
    static uint64_t cached_hash = 0;
    static uint32_t cached_n = 0;
    obvh *accel = NULL;
    if (n > OBVH_MIN_SPHERES && !getenv("ORACLE_NO_BVH")) {
        uint64_t hsh = 1469598103934665603ull;
        for (uint32_t i = 0; i < n; i++) { /* PtSphere = 16 bytes = two 64-bit words */
            uint64_t w64[2];
            memcpy(w64, &spheres[i], sizeof w64);
            hsh = (hsh ^ w64[0]) * 1099511628211ull;
            hsh = (hsh ^ w64[1]) * 1099511628211ull;
        }
        if (!cached || cached_n != n || cached_hash != hsh) {
            obvh_free(cached);
            cached = obvh_build(spheres, n);
            cached_n = n; cached_hash = hsh;
        }
        accel = cached;
    }
    /* emitters in id order (any emission component > 0) */
    uint32_t *light_ids = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    light_list ll = { light_ids, 0 };
    for (uint32_t i = 0; i < n; i++) {
        const PtMaterial *m = &materials[i];
        if (m->EmissiveStrength * m->EmissiveColor[0] > 0.0f || m->EmissiveStrength * m->EmissiveColor[1] > 0.0f || m->EmissiveStrength * m->EmissiveColor[2] > 0.0f)
            light_ids[ll.n++] = i;
    }
    int has_alpha = 0;
    for (uint32_t i = 0; i < n && !has_alpha; i++) has_alpha = materials[i].AlphaMode != PT_ALPHA_OPAQUE;
    if (row_step == 0) row_step = 1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        job_t *j = &jobs[t];
        j->sph = spheres; j->mat = materials; j->n = n; j->sd = scene; j->cam = camera; j->gs = gs;
        j->rect = *rect; j->row_step = row_step; j->out = out_rgba; j->tid = t; j->nthreads = threads;
        j->tex = textured ? &tc : NULL;
        j->lights = &ll;
        j->accel = accel;
        j->has_alpha = has_alpha;
    }
    if (threads == 1) worker(&jobs[0]);
    else {
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    if (stats) {
        stats->rays = 0; stats->paths = 0;
        for (int t = 0; t < threads; t++) { stats->rays += jobs[t].rays; stats->paths += jobs[t].paths; }
    }
    free(jobs); free(th); free(light_ids);
    return 0;
}

int oracle_trace_pixel(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                       const PtSceneData *scene, const PtCamera *camera,
                       const PtGraphicsSettings *gs, uint32_t px, uint32_t py,
                       float *events, uint32_t max_events, uint32_t *n_events)
{
    int err = validate(scene, gs, n);
    if (err) return err;
    trace_t tr = { events, max_events, 0 };
    float rgba[4]; uint64_t paths = 0;
    int has_alpha = 0;
    for (uint32_t i = 0; i < n && !has_alpha; i++) has_alpha = materials[i].AlphaMode != PT_ALPHA_OPAQUE;
    render_pixel(spheres, materials, n, scene, camera, gs, px, py, rgba, &paths, &tr, NULL, NULL, NULL, has_alpha);
    *n_events = tr.n_events;
    return 0;
}
