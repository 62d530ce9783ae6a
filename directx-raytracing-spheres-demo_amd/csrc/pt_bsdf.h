// pt_bsdf.h -- the 3-lobe metallic/roughness BSDF of Shaders/BxDF.hlsli and the pinhole camera of
// Shaders/Camera.hlsli, as device functions (also host-compilable for the leaf parity tests).
#pragma once

#include "pt_math.h"
#include "../../include/pt_types.h"

namespace pt {

enum : int { kLobeDiffuse = 0, kLobeSpecular = 1, kLobeTransmission = 2 };

// BSDFSample (BxDF.hlsli:36-79)
struct Bsdf {
    f3 BaseColor;
    float Metallic;
    f3 Albedo;
    float Roughness, IORi, IORo, Eta;
    f3 F0;
    float Transmission;
};

// F0 of the dielectric interface: ((IORi - IORo) / (IORi + IORo))^2 is the same for both orientations (the quotient only
// changes sign), so it is a per-material constant; pt_set_scene stores it (and 1/IOR) in the material's padding words.
PT_HD float dielectric_f0(float ior) { float r = (1.0f - ior) / (1.0f + ior); return r * r; }

// f0d = dielectric_f0(ior) and inv_ior = 1/ior may be precomputed per material (bit-identical to computing them here)
PT_HD Bsdf bsdf_init_pre(f3 base, float metallic, float roughness, float ior, float inv_ior, float f0d, float transmission, bool front)
{
    Bsdf b;
    b.BaseColor = base;
    b.Metallic = metallic;
    b.Albedo = base * (1.0f - metallic);
    b.Roughness = pt_max(kMinRoughness, roughness);
    b.IORi = front ? 1.0f : ior;
    b.IORo = front ? ior : 1.0f;
    b.Eta = front ? inv_ior : ior;  // IORi / IORo: 1/ior, or ior/1 = ior exactly
    // lerp(f0d, baseColor, metallic), f0d = pow((IORi - IORo) / (IORi + IORo), 2) of BxDF.hlsli:64 restated as x*x
    b.F0 = make_f3(pt_fma(metallic, base.x - f0d, f0d), pt_fma(metallic, base.y - f0d, f0d), pt_fma(metallic, base.z - f0d, f0d));
    b.Transmission = transmission;
    return b;
}

PT_HD Bsdf bsdf_init(f3 base, float metallic, float roughness, float ior, float transmission, bool front)
{
    return bsdf_init_pre(base, metallic, roughness, ior, 1.0f / ior, dielectric_f0(ior), transmission, front);
}

// SurfaceVectors (SurfaceVectors.hlsli:5-15)
struct Surf {
    f3 FrontNg, Ns;
    Basis basis;
};

PT_HD Surf surf_init(bool front, f3 Ng, f3 Ns)
{
    Surf s;
    s.FrontNg = front ? Ng : -Ng;
    s.Ns = Ns;
    s.basis = get_basis(Ns);
    return s;
}

// EstimateDiffuseProbability (BxDF.hlsli:21-34)
PT_HD float estimate_diffuse_probability(f3 albedo, f3 f0, float roughness, float nov)
{
    f3 fe = environment_term_rtg(f0, nov, roughness);
    float diffuse = luminance(albedo * make_f3(1.0f - fe.x, 1.0f - fe.y, 1.0f - fe.z));
    float specular = luminance(fe);
    float sum = diffuse + specular;
    float p = sum > 0.0f ? diffuse / sum : 1.0f;
    if (0.0f < p && p < 1.0f) p = p < 0.05f ? 0.05f : (p > 0.95f ? 0.95f : p);
    return p;
}

// ComputeLobeWeights (BxDF.hlsli:184-196)
PT_HD void lobe_weights(const Bsdf& b, const Surf& s, f3 V, float w[3])
{
    float nov = pt_abs(dot(s.Ns, V));
    float wt = b.Transmission * (1.0f - b.Metallic);
    float wr = 1.0f - wt;
    float pd = estimate_diffuse_probability(b.Albedo, b.F0, b.Roughness, nov);
    float ps = 1.0f - pd;
    w[kLobeDiffuse] = pd * wr;
    w[kLobeSpecular] = ps * wr;
    w[kLobeTransmission] = wt;
}

// FindLobe (BxDF.hlsli:198-212)
PT_HD int find_lobe(const float w[3], float rnd)
{
    float weight = 0.0f;
    weight += w[2];
    if (rnd < weight) return 2;
    weight += w[1];
    if (rnd < weight) return 1;
    return 0;
}

// Sample (BxDF.hlsli:214-226; :81-86, :110-118, :148-170)
PT_HD bool bsdf_sample(const Bsdf& b, const Surf& s, f3 V, const float w[3], const float rnd[4], f3& L, int& lobe)
{
    lobe = find_lobe(w, rnd[0]);
    if (lobe == kLobeDiffuse) {
        L = rotate_vector_inverse(s.basis, cosine_ray(rnd[1], rnd[2]));
        return dot(s.FrontNg, L) > 0.0f;
    }
    f3 Vl = rotate_vector(s.basis, V);
    f3 H = rotate_vector_inverse(s.basis, vndf_ray(rnd[1], rnd[2], b.Roughness, Vl));
    if (lobe == kLobeSpecular) {
        L = reflect(-V, H);
        return dot(s.FrontNg, L) > 0.0f;
    }
    float voh = pt_abs(dot(V, H));
    float eta = b.Eta;
    if (eta * eta * (1.0f - voh * voh) > 1.0f || rnd[3] < fresnel_dielectric(eta, voh)) {
        L = reflect(-V, H);
    } else {
        L = refract(-V, H, eta);
        if (!is_finite(L.x) || !is_finite(L.y) || !is_finite(L.z)) L = -V;
    }
    return true;
}

// ComputeHalfVector (BxDF.hlsli:228-245)
PT_HD f3 half_vector(const Bsdf& b, const Surf& s, f3 L, f3 V, bool transmissive)
{
    f3 N = s.FrontNg;
    f3 H;
    if (transmissive && dot(N, L) < 0.0f) {
        H = normalize(mad(b.IORo, L, V * b.IORi));
        if (dot(N, H) < 0.0f) H = -H;
    } else {
        H = normalize(L + V);
    }
    return H;
}

// EvaluatePDF(lobeType) (BxDF.hlsli:287-299)
PT_HD float bsdf_pdf(const Bsdf& b, const Surf& s, f3 L, f3 V, const float w[3], int lobe)
{
    f3 H = half_vector(b, s, L, V, w[kLobeTransmission] > 0.0f);
    float lw = w[lobe];
    f3 N = s.Ns;
    if (lobe == kLobeDiffuse) {
        if (dot(s.FrontNg, L) > 0.0f) { float nol = pt_abs(dot(N, L)); return (nol * kInvPi) * lw; }
        return 0.0f * lw;
    }
    if (lobe == kLobeSpecular) {
        if (dot(s.FrontNg, L) > 0.0f) {
            f3 Vl = rotate_vector(s.basis, V);
            float noh = pt_abs(dot(N, H));
            return vndf_pdf(Vl, noh, b.Roughness) * lw;
        }
        return 0.0f * lw;
    }
    return pt_abs(dot(N, L)) * lw;
}

// Evaluate(lobeType) (BxDF.hlsli:301-315)
PT_HD f3 bsdf_eval(const Bsdf& b, const Surf& s, f3 L, f3 V, const float w[3], int lobe)
{
    float wt = w[kLobeTransmission];
    f3 H = half_vector(b, s, L, V, wt > 0.0f);
    f3 N = s.Ns;
    if (lobe == kLobeTransmission) {
        float nol = pt_abs(dot(N, L));
        return (b.BaseColor * nol) * wt;
    }
    float wr = 1.0f - wt;
    if (!(dot(s.FrontNg, L) > 0.0f)) return make_f3(0.0f, 0.0f, 0.0f) * wr;
    float nol = pt_abs(dot(N, L)), nov = pt_abs(dot(N, V)), voh = pt_abs(dot(V, H));
    if (lobe == kLobeDiffuse) {
        float dt = diffuse_term(b.Roughness, nol, nov, voh);
        return ((b.Albedo * nol) * dt) * wr;
    }
    float noh = pt_abs(dot(N, H));
    float D = distribution_term(b.Roughness, noh);
    float G = geometry_term_mod(b.Roughness, nol, nov);
    f3 F = fresnel_schlick(b.F0, voh);
    float k = nol * D * G;
    return (F * k) * wr;
}

// EvaluatePDF(lobeType) followed by Evaluate(lobeType) of the lobe that was sampled (Raytracing.hlsl:335-345), with what the two share
// computed once: the half vector (one normalisation instead of two), NoL, and for the specular lobe Vlocal and NoH.  Every value is
// formed by the very operations of bsdf_pdf / bsdf_eval above, so pdf and f are bit-identical to theirs (tests/test_leaf_parity.py).
// Returns false when pdf == 0 (the caller ends the sample, :336-339); f is then not evaluated.
PT_HD bool bsdf_pdf_eval(const Bsdf& b, const Surf& s, f3 L, f3 V, const float w[3], int lobe, float& pdf, f3& f)
{
    const float wt = w[kLobeTransmission];
    const float lw = w[lobe];
    const f3 N = s.Ns;
    const float nol = pt_abs(dot(N, L));
    f = make_f3(0.0f, 0.0f, 0.0f);
    if (lobe == kLobeTransmission) {
        pdf = nol * lw;
        if (pdf == 0.0f) return false;
        f = (b.BaseColor * nol) * wt;
        return true;
    }
    const bool front = dot(s.FrontNg, L) > 0.0f;
    const float wr = 1.0f - wt;
    if (lobe == kLobeDiffuse) {
        pdf = front ? (nol * kInvPi) * lw : 0.0f * lw;
        if (pdf == 0.0f) return false;
        if (!front) { f = make_f3(0.0f, 0.0f, 0.0f) * wr; return true; }  // (pdf is NaN here: lw is not finite)
        const f3 H = half_vector(b, s, L, V, wt > 0.0f);
        const float nov = pt_abs(dot(N, V)), voh = pt_abs(dot(V, H));
        const float dt = diffuse_term(b.Roughness, nol, nov, voh);
        f = ((b.Albedo * nol) * dt) * wr;
        return true;
    }
    // specular reflection
    if (!front) {
        pdf = 0.0f * lw;
        if (pdf == 0.0f) return false;
        f = make_f3(0.0f, 0.0f, 0.0f) * wr;
        return true;
    }
    const f3 H = half_vector(b, s, L, V, wt > 0.0f);
    const f3 Vl = rotate_vector(s.basis, V);
    const float noh = pt_abs(dot(N, H));
    pdf = vndf_pdf(Vl, noh, b.Roughness) * lw;
    if (pdf == 0.0f) return false;
    const float nov = pt_abs(dot(N, V)), voh = pt_abs(dot(V, H));
    const float D = distribution_term(b.Roughness, noh);
    const float G = geometry_term_mod(b.Roughness, nol, nov);
    const f3 F = fresnel_schlick(b.F0, voh);
    const float k = nol * D * G;
    f = (F * k) * wr;
    return true;
}

// Camera::GeneratePinholeRay (Camera.hlsli:27-41) with Math::CalculateUV/NDC (Math.hlsli:7-15)
struct CameraParams {
    f3 Position, Right, Up, Forward;
    float Near, Far, JitterX, JitterY;
    // per-frame constants hoisted out of the per-pixel code (computed once on the host with the same arithmetic)
    f3 ForwardN;        // normalize(Forward) of GeneratePinholeRay
    float InvW, InvH;   // 1 / RenderSize of CalculateUV
};

PT_HD CameraParams camera_params(const PtCamera& c, uint32_t w, uint32_t h)
{
    CameraParams p;
    p.Position = make_f3(c.Position[0], c.Position[1], c.Position[2]);
    p.Right = make_f3(c.RightDirection[0], c.RightDirection[1], c.RightDirection[2]);
    p.Up = make_f3(c.UpDirection[0], c.UpDirection[1], c.UpDirection[2]);
    p.Forward = make_f3(c.ForwardDirection[0], c.ForwardDirection[1], c.ForwardDirection[2]);
    p.Near = c.NearDepth; p.Far = c.FarDepth; p.JitterX = c.Jitter[0]; p.JitterY = c.Jitter[1];
    p.ForwardN = normalize(p.Forward);
    p.InvW = 1.0f / (float)w;
    p.InvH = 1.0f / (float)h;
    return p;
}

PT_HD void primary_ray(const CameraParams& cam, uint32_t px, uint32_t py, f3& o, f3& d, float& tmin, float& tmax)
{
    float u = ((float)px + 0.5f + cam.JitterX) * cam.InvW;
    float v = ((float)py + 0.5f + cam.JitterY) * cam.InvH;
    float nx = pt_fma(u, 2.0f, -1.0f);
    float ny = pt_fma(v, -2.0f, 1.0f);
    f3 dir = mad(ny, cam.Up, cam.Right * nx) + cam.Forward;
    dir = normalize(dir);
    float inv_cos = pt_rcp(dot(cam.ForwardN, dir));
    o = cam.Position;
    d = dir;
    tmin = cam.Near * inv_cos;
    tmax = cam.Far * inv_cos;
}

}  // namespace pt
