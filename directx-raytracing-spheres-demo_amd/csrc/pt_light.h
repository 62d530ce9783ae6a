// pt_light.h -- sphere-light sampling for the direct-illumination pass: a functional stand-in for the
// reference's RTXDI / ReSTIR-DI passes (LightPreparation.ixx:52-70 enumerates the emissive meshes, DIInitialSampling.hlsl ...
// DIFinalShading.hlsl produce the DI texture Raytracing.hlsl:150-163 reads).  One emitter is chosen uniformly, a direction
// uniformly inside the cone the emitter subtends, and the estimate is Le * f(L) cos / pdf if the emitter is the first thing
// the shadow ray meets.  Device functions that also compile on the host for the leaf parity tests.
#pragma once

#include "pt_bsdf.h"

namespace pt {

struct LightSample {
    f3 L;           // unit direction towards the emitter
    float inv_pdf;  // 1 / pdf of L with respect to solid angle = the cone's solid angle
    bool valid;     // false: the shading point is inside (or on) the emitter
};

// Uniform direction inside the cone that the sphere (C, r) subtends from P; u1, u2 in (0, 1].
// 1 - cos(theta_max) is formed as sin^2 / (1 + cos) so that small distant emitters keep their solid angle.
PT_HD LightSample sample_sphere_cone(f3 P, f3 C, float r, float u1, float u2)
{
    LightSample s;
    s.L = make_f3(0.0f, 0.0f, 1.0f); s.inv_pdf = 0.0f; s.valid = false;
    const f3 w = C - P;
    const float d2 = dot(w, w), r2 = r * r;
    if (!(d2 > r2)) return s;
    const f3 wn = w * pt_rcp(pt_sqrt(d2));
    const float sin2 = r2 / d2;
    const float cos_max = sqrt01(1.0f - sin2);
    const float omc = sin2 / (1.0f + cos_max);       // 1 - cos(theta_max)
    const float k = u1 * omc;                        // 1 - cos(theta)
    const float cos_t = 1.0f - k;
    const float sin_t = sqrt01(k * (1.0f + cos_t));  // sin^2 = (1 - cos)(1 + cos)
    float sp, cp;
    sincos_2pi(u2, sp, cp);
    const Basis b = get_basis(wn);
    s.L = rotate_vector_inverse(b, make_f3(sin_t * cp, sin_t * sp, cos_t));
    s.inv_pdf = 6.28318530717958647692f * omc;
    s.valid = true;
    return s;
}

// index of the chosen emitter among n_lights, u in (0, 1]
PT_HD uint32_t pick_light(float u, uint32_t n_lights)
{
    const uint32_t j = (uint32_t)(u * (float)n_lights);
    return j < n_lights ? j : n_lights - 1u;
}

// reflective part of the BSDF times cos for an arbitrary direction L: diffuse + specular lobes, each with the (1 - wT) factor
// Evaluate gives it (BxDF.hlsli:301-315); zero below the geometric horizon
PT_HD f3 bsdf_eval_reflective(const Bsdf& b, const Surf& s, f3 L, f3 V, const float w[3])
{
    return bsdf_eval(b, s, L, V, w, kLobeDiffuse) + bsdf_eval(b, s, L, V, w, kLobeSpecular);
}

constexpr float kDiNegligible = 1e-7f;  // upper bound of an estimate below which no shadow ray is cast (di_estimate)
constexpr uint32_t kDiRngSalt = 0x44495F31u;  // the DI pass has its own per-pixel stream: rng_init(px, py, FrameIndex ^ salt)

}  // namespace pt
