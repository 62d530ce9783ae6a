// pt_post.h -- display transform and progressive accumulation, as device functions that also compile
// on the host for the leaf parity tests.
//
// Replaces App::Impl::ToneMap (Source/App.cpp:1731-1757), which drives DirectXTK's ToneMapPostProcess (un-vendored, vcpkg
// `directxtk12`): SDR = operator {Saturate, Reinhard, ACESFilmic} on `hdr * 2^Exposure` followed by the sRGB estimate
// pow(|c|, 1/2.2); HDR10 = no operator, colour-primary rotation, `* PaperWhiteNits / 10000`, SMPTE ST 2084 OETF
// (Source/App.cpp:760-769, 1740-1748).  The shader bodies are restated from the published DirectXTK ToneMap.fx
// (recollection; build-frozen like the rest of the un-vendored arithmetic); pow is pow_spec, so the
// GPU and the CPU oracle agree bit for bit.
#pragma once

#include "pt_math.h"
#include "../../include/pt_types.h"

namespace pt {

enum : uint32_t { kToneNone = 0, kToneSaturate = 1, kToneReinhard = 2, kToneACESFilmic = 3 };  // ToneMapPostProcess::Operator
enum : uint32_t { kTransferLinear = 0, kTransferSRGB = 1, kTransferST2084 = 2 };               // ::TransferFunction
enum : uint32_t { kRotate709to2020 = 0, kRotateP3D65to2020 = 1, kRotate709toP3D65 = 2 };       // ::ColorPrimaryRotation

// pow for the display curves: x is a non-negative finite value; pow(0, y) = 0 (pow_spec's log2 needs x > 0)
PT_HD float pow_pos(float x, float y) { return x > 0.0f ? pow_spec(x, y) : 0.0f; }

PT_HD float tone_operator(float x, uint32_t op)
{
    if (op == kToneSaturate) return saturate(x);
    if (op == kToneReinhard) return x / (1.0f + x);
    if (op == kToneACESFilmic) {
        // Narkowicz fit: saturate((x (a x + b)) / (x (c x + d) + e))
        const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
        return saturate((x * pt_fma(a, x, b)) / pt_fma(x, pt_fma(c, x, d), e));
    }
    return x;
}

// LinearToSRGBEst of ToneMap.fx: pow(abs(c), 1 / 2.2); the input is clamped to [0, 1] first (what the UNORM target does
// to the output anyway; it also keeps NaN / inf away from pow)
PT_HD float linear_to_srgb_est(float c) { return pow_pos(saturate(c), 1.0f / 2.2f); }

// LinearToST2084: ((c1 + c2 Y^m1) / (1 + c3 Y^m1))^m2
PT_HD float linear_to_st2084(float y)
{
    const float ym = pow_pos(pt_min(pt_abs(y), 1.0e30f), 0.1593017578f);
    return pow_pos(pt_fma(18.8515625f, ym, 0.8359375f) / pt_fma(18.6875f, ym, 1.0f), 78.84375f);
}

PT_HD f3 rotate_primaries(f3 c, uint32_t rotation)
{
    if (rotation == kRotateP3D65to2020)
        return make_f3(dot(make_f3(0.753845f, 0.198593f, 0.047562f), c), dot(make_f3(0.0457456f, 0.941777f, 0.0124772f), c),
                       dot(make_f3(-0.00121055f, 0.0176041f, 0.983607f), c));
    if (rotation == kRotate709toP3D65)
        return make_f3(dot(make_f3(0.822461969f, 0.1775380f, 0.0f), c), dot(make_f3(0.033194199f, 0.966805801f, 0.0f), c),
                       dot(make_f3(0.017082631f, 0.0723974f, 0.910519969f), c));
    return make_f3(dot(make_f3(0.6274040f, 0.3292820f, 0.0433136f), c), dot(make_f3(0.0690970f, 0.9195400f, 0.0113612f), c),
                   dot(make_f3(0.0163916f, 0.0880132f, 0.8955950f), c));
}

PT_HD uint32_t unorm(float v, float scale) { return (uint32_t)pt_fma(saturate(v), scale, 0.5f); }  // NaN -> 0 (D3D conversion rule)

// One pixel of ToneMapPostProcess::Process.  SDR (Linear / SRGB): R8G8B8A8_UNORM, alpha 255.  ST2084: R10G10B10A2_UNORM, alpha 3.
PT_HD uint32_t tonemap_pixel(f3 hdr, const PtToneMapParams& p)
{
    if (p.TransferFunction == kTransferST2084) {
        const f3 r = rotate_primaries(hdr, p.ColorRotation);
        const float k = p.PaperWhiteNits * (1.0f / 10000.0f);
        const float x = linear_to_st2084(r.x * k), y = linear_to_st2084(r.y * k), z = linear_to_st2084(r.z * k);
        return unorm(x, 1023.0f) | (unorm(y, 1023.0f) << 10) | (unorm(z, 1023.0f) << 20) | (3u << 30);
    }
    float x = tone_operator(hdr.x * p.LinearExposure, p.Operator);
    float y = tone_operator(hdr.y * p.LinearExposure, p.Operator);
    float z = tone_operator(hdr.z * p.LinearExposure, p.Operator);
    if (p.TransferFunction == kTransferSRGB) { x = linear_to_srgb_est(x); y = linear_to_srgb_est(y); z = linear_to_srgb_est(z); }
    return unorm(x, 255.0f) | (unorm(y, 255.0f) << 8) | (unorm(z, 255.0f) << 16) | (255u << 24);
}

// Progressive accumulation: running mean over frames, accum_{n+1} = accum_n + (x - accum_n) / (n + 1); n = frames already
// accumulated (n = 0 stores x).  `inv` = 1 / (float)(n + 1), computed once by the caller.
PT_HD float accumulate_value(float accum, float x, float inv, bool first) { return first ? x : pt_fma(x - accum, inv, accum); }

}  // namespace pt
