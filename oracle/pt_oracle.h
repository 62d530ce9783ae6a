/* pt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar plain-C restatement of the reference's per-pixel Monte-Carlo bounce loop
 * (Shaders/Raytracing.hlsl:103-415, DEFAULT permutation, IsDIEnabled = 0,
 * Denoiser::None) with the primary-hit pass (Shaders/GBufferGeneration.hlsl:117-232)
 * folded in, over analytic spheres with BRUTE-FORCE O(N) intersection.
 *
 * PARITY UNPINNED: the reference has no tests/golden vectors and its
 * arithmetic core (NVIDIA MathLib ml.hlsli) is an un-vendored submodule whose pinned
 * commit is unknown.  MathLib functions are restated from SURVEY
 * Appendix A (the build-frozen spec); see DESIGN.md "Frozen arithmetic spec".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (HIP) path never links or calls it.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleStats {
    uint64_t rays;   /* CastRay invocations incl. primaries */
    uint64_t paths;  /* (pixel, sample) pairs started */
} OracleStats;

/* Render rect (w x h pixels, every row_step-th row of the rect is rendered, the others
 * are left untouched) of the RenderSize image into out_rgba (rect.w*rect.h*4 floats,
 * row-major inside the rect).  threads <= 1: single thread.  Returns 0 on success. */
int oracle_render(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                  const PtSceneData *scene, const PtCamera *camera,
                  const PtGraphicsSettings *gs, const PtRect *rect, uint32_t row_step,
                  float *out_rgba, OracleStats *stats, int threads);

/* Row N1: the same render with textured spheres (EvaluateMaterial's texture branches, Shaders/ShadingHelpers.hlsli:53-103,
 * 161-235).  `textures` may be NULL (= oracle_render). */
typedef struct OracleTextures {
    const PtTexture *textures;               /* decoded images; SceneData.EnvironmentLightTextureDescriptor indexes this table too */
    uint32_t n_textures;
    const PtObjectTextures *object_textures; /* one TextureMapInfoArray per sphere, or NULL = no sphere has maps */
    const float *rotations;                  /* n quaternions (x, y, z, w) object -> world, or NULL = identity */
} OracleTextures;
int oracle_render_textured(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                           const PtSceneData *scene, const PtCamera *camera,
                           const PtGraphicsSettings *gs, const PtRect *rect, uint32_t row_step,
                           float *out_rgba, OracleStats *stats, int threads, const OracleTextures *textures);
/* Closest hit of one ray: brute force (use_bvh = 0, the definition) or the oracle's own median-split BVH, which must agree
 * exactly.  bvh_cache (may be NULL): *bvh_cache keeps the built structure between calls; release with oracle_free_bvh. */
int oracle_closest_hit(const PtSphere *spheres, uint32_t n, const float o[3], const float d[3], float tmin, float tmax, int use_bvh,
                       void **bvh_cache, float *t, uint32_t *id);
void oracle_free_bvh(void *bvh);
/* ... with alpha-tested hits (spec S10; RaytracingHelpers.hlsli:19-43, ShadingHelpers.hlsli:105-115): brute force, textures may be NULL */
int oracle_closest_hit_alpha(const PtSphere *spheres, const PtMaterial *materials, uint32_t n, const OracleTextures *textures,
                             const float o[3], const float d[3], float tmin, float tmax, float *t, uint32_t *id);
/* leaf of row N4: uniform direction in the cone the sphere (C, r) subtends from P; returns 0 when P is inside the sphere */
int oracle_sample_sphere_cone(const float P[3], const float C[3], float r, float u1, float u2, float L[3], float *inv_pdf);
/* leaves of row N1 */
float oracle_atan2(float y, float x);
void oracle_sphere_uv(const float n[3], float uv[2]);
void oracle_latlong_uv(const float d[3], float uv[2]); /* Math::ToLatLongCoordinate (Math.hlsli:29-33), row a18's texture branch */
uint32_t oracle_cube_face_uv(const float d[3], float uv[2]); /* TextureCube face selection + face coordinates (spec S9); returns the face */
void oracle_sphere_tangent(const float n[3], float t[3]);
void oracle_quat_rotate(const float q[4], const float v[3], float out[3]);
void oracle_sample_texture(const OracleTextures *t, uint32_t index, const float uv[2], float out[4]);
void oracle_perturb_normal(const float N[3], const float T[3], float sx, float sy, float out[3]);

/* Per-bounce trace of one pixel, for debugging parity: each event is 16 floats
 * {sample, bounce, hit_id(as float bits), t, Px,Py,Pz, Lx,Ly,Lz, Tr,Tg,Tb, rng_state(bits), lobe, flags}. */
int oracle_trace_pixel(const PtSphere *spheres, const PtMaterial *materials, uint32_t n,
                       const PtSceneData *scene, const PtCamera *camera,
                       const PtGraphicsSettings *gs, uint32_t px, uint32_t py,
                       float *events, uint32_t max_events, uint32_t *n_events);

/* ---- leaf functions exported for known-answer tests ---- */
uint32_t oracle_hash(uint32_t x);
uint32_t oracle_rng_init(uint32_t px, uint32_t py, uint32_t frame);
uint32_t oracle_rng_next(uint32_t *state);            /* GetUint */
float oracle_rng_float(uint32_t *state);               /* GetFloat, (0,1] */
float oracle_halton(uint32_t index, uint32_t base);
void oracle_sincos_2pi(float u, float *s, float *c);
float oracle_log2(float x);
float oracle_exp2(float x);
float oracle_pow(float x, float y);
float oracle_from_srgb(float c);

/* row N3: display transform (Source/App.cpp:1731-1757 -> DirectXTK ToneMapPostProcess) and progressive accumulation */
uint32_t oracle_tonemap_pixel(const float hdr[3], const PtToneMapParams *params);
void oracle_tonemap(const float *hdr_rgba, uint32_t n_pixels, const PtToneMapParams *params, uint32_t *out);
void oracle_accumulate(float *accum_rgba, const float *radiance_rgba, uint32_t n_pixels, uint32_t frames_accumulated);
void oracle_get_basis(const float n[3], float t[3], float b[3]);
void oracle_cosine_ray(const float u[2], float out[3]);
void oracle_vndf_ray(const float u[2], float roughness, const float vlocal[3], float out[3]);
float oracle_vndf_pdf(const float vlocal[3], float noh, float roughness);
float oracle_distribution_term(float roughness, float noh);
float oracle_geometry_term_mod(float roughness, float nol, float nov);
float oracle_fresnel_dielectric(float eta, float von);
float oracle_diffuse_term(float roughness, float nol, float nov, float voh);
void oracle_environment_term_rtg(const float f0[3], float nov, float roughness, float out[3]);
void oracle_sky(const PtSceneData *scene, const float dir[3], float out[3]);
/* returns 1 on hit and writes t */
int oracle_intersect_sphere(const float o[3], const float d[3], float tmin, float tmax,
                            const PtSphere *s, float *t);
/* in: ray, t, sphere; out: P (re-projected), N (outward), offset, front */
void oracle_hit_frame(const float o[3], const float d[3], float t, const PtSphere *s,
                      float P[3], float N[3], float *offset, int *front);
void oracle_spawn_origin(const float P[3], const float N[3], float offset, const float L[3], float out[3]);
void oracle_primary_ray(const PtCamera *cam, uint32_t px, uint32_t py, uint32_t w, uint32_t h,
                        float o[3], float d[3], float *tmin, float *tmax);

/* One BSDF interaction (BxDF.hlsli): given material, front flag, outward normal N
 * (geometric), view V and the 4 random numbers, compute lobe, L, pdf and f.
 * returns 1 if a direction was sampled (Sample() returned true). */
typedef struct OracleBsdfOut {
    int lobe;
    int valid;
    float L[3];
    float pdf;
    float f[3];
    float weights[3];
} OracleBsdfOut;
void oracle_bsdf_step(const PtMaterial *m, int front, const float Ng[3], const float V[3],
                      const float rnd[4], OracleBsdfOut *out);

#ifdef __cplusplus
}
#endif
#endif
