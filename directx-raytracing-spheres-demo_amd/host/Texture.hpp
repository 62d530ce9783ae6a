// Texture.hpp -- host mirror of the texture side of the scene (row N1): TextureMapType / TextureMapInfo
// (Source/Material.ixx:22-38), the decoded image a loader hands over (Source/TextureHelpers.ixx:34-60: 8-bit RGBA,
// optionally forced to an sRGB format), and a loader hook.  The reference loads Assets/Textures/*.png|*.jpg through
// DirectXTex; those assets do not travel with this repository, so the default loader returns deterministic procedural
// stand-ins for the file names the demo scene asks for (Source/MyScene.ixx:161-166, 277-295).  Install a real loader with
// Scene::SetTextureLoader to use decoded files.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pt_types.h"
#include "Material.hpp"  // TextureMapType, TextureMapInfo
#include "Random.hpp"

namespace dxrs {

struct Texture {
    uint32_t Width = 0, Height = 0;
    bool ForceSRGB = false;           // colour data: sampled as DXGI_FORMAT_R8G8B8A8_UNORM_SRGB
    std::vector<uint8_t> Pixels;      // Width * Height * 4, row-major RGBA
    std::vector<float> HDRPixels;     // or: Width * Height * 4 linear floats (an .exr / .hdr environment map); Pixels empty
    std::vector<Texture> Faces;       // or: the six square faces of a cube map (+X, -X, +Y, -Y, +Z, -Z), this entry itself empty

    bool IsCubeMap() const noexcept { return Faces.size() == 6; }  // Texture::IsCubeMap (Source/App.cpp:984)

    bool IsHDR() const noexcept { return !HDRPixels.empty(); }
    uint32_t Format() const noexcept { return IsHDR() ? PT_TEXTURE_RGBA32_FLOAT : ForceSRGB ? PT_TEXTURE_RGBA8_UNORM_SRGB : PT_TEXTURE_RGBA8_UNORM; }
    size_t ByteSize() const noexcept { return IsHDR() ? HDRPixels.size() * sizeof(float) : Pixels.size(); }
    const void* Data() const noexcept { return IsHDR() ? static_cast<const void*>(HDRPixels.data()) : Pixels.data(); }

    PtTexture ToPt() const
    {
        PtTexture t{};
        t.Pixels = Data(); t.Width = Width; t.Height = Height; t.Format = Format();
        return t;
    }
};

// textureMapType: a TextureMapType, or EnvironmentLightTexture for SceneDesc::EnvironmentLight.Texture (Scene.ixx:130-133)
constexpr uint32_t EnvironmentLightTexture = TextureMapType::Count;
using TextureLoader = std::function<Texture(const std::string& path, uint32_t textureMapType)>;

namespace procedural {

// tileable fractal value noise in [0, 1]
inline std::vector<float> ValueNoise(uint32_t w, uint32_t h, unsigned seed, int octaves = 5)
{
    Random random(seed);
    std::vector<float> out(static_cast<size_t>(w) * h, 0.0f);
    float amplitude = 1, total = 0;
    for (int o = 0; o < octaves; o++) {
        const uint32_t n = 4u << o;
        std::vector<float> grid(static_cast<size_t>(n) * n);
        for (auto& g : grid) g = random.Float();
        for (uint32_t y = 0; y < h; y++) {
            const float fy0 = static_cast<float>(y) * n / h;
            const uint32_t y0 = static_cast<uint32_t>(fy0);
            float fy = fy0 - y0; fy = fy * fy * (3 - 2 * fy);
            for (uint32_t x = 0; x < w; x++) {
                const float fx0 = static_cast<float>(x) * n / w;
                const uint32_t x0 = static_cast<uint32_t>(fx0);
                float fx = fx0 - x0; fx = fx * fx * (3 - 2 * fx);
                const float g00 = grid[(y0 % n) * n + x0 % n], g10 = grid[(y0 % n) * n + (x0 + 1) % n];
                const float g01 = grid[((y0 + 1) % n) * n + x0 % n], g11 = grid[((y0 + 1) % n) * n + (x0 + 1) % n];
                out[static_cast<size_t>(y) * w + x] += amplitude * ((g00 * (1 - fx) + g10 * fx) * (1 - fy) + (g01 * (1 - fx) + g11 * fx) * fy);
            }
        }
        total += amplitude;
        amplitude *= 0.5f;
    }
    for (auto& v : out) v /= total;
    return out;
}

inline uint8_t ToByte(float v) { return static_cast<uint8_t>(std::lround(std::fmin(std::fmax(v, 0.0f), 1.0f) * 255.0f)); }

// two-colour map driven by a height field (continents / maria / oxide patches)
inline Texture Albedo(uint32_t w, uint32_t h, unsigned seed, const float low[3], const float high[3], float level)
{
    const auto n = ValueNoise(w, h, seed);
    Texture t; t.Width = w; t.Height = h; t.ForceSRGB = true; t.Pixels.resize(n.size() * 4);
    for (size_t i = 0; i < n.size(); i++) {
        const float* c = n[i] > level ? high : low;
        const float shade = 0.6f + 0.8f * n[i];
        for (int k = 0; k < 3; k++) t.Pixels[4 * i + k] = ToByte(c[k] * shade);
        t.Pixels[4 * i + 3] = 255;
    }
    return t;
}

inline Texture Scalar(uint32_t w, uint32_t h, unsigned seed, float lo, float hi)
{
    const auto n = ValueNoise(w, h, seed);
    Texture t; t.Width = w; t.Height = h; t.Pixels.resize(n.size() * 4);
    for (size_t i = 0; i < n.size(); i++) {
        const uint8_t v = ToByte(lo + (hi - lo) * n[i]);
        t.Pixels[4 * i] = t.Pixels[4 * i + 1] = t.Pixels[4 * i + 2] = v; t.Pixels[4 * i + 3] = 255;
    }
    return t;
}

// tangent-space normal map from a height field, xy in RG as Geometry::UnpackLocalNormal decodes it: s = (n + 1) * 127 / 255
inline Texture NormalMap(uint32_t w, uint32_t h, unsigned seed, float strength)
{
    const auto n = ValueNoise(w, h, seed);
    Texture t; t.Width = w; t.Height = h; t.Pixels.resize(n.size() * 4);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const auto at = [&](uint32_t xx, uint32_t yy) { return n[static_cast<size_t>(yy % h) * w + xx % w]; };
            const float dx = (at(x + 1, y) - at(x + w - 1, y)) * 0.5f * strength * w / 64;
            const float dy = (at(x, y + 1) - at(x, y + h - 1)) * 0.5f * strength * h / 64;
            const float inv = 1.0f / std::sqrt(dx * dx + dy * dy + 1);
            const size_t i = static_cast<size_t>(y) * w + x;
            t.Pixels[4 * i] = static_cast<uint8_t>(std::lround(std::fmin(std::fmax((-dx * inv + 1) * 127.0f, 0.0f), 254.0f)));
            t.Pixels[4 * i + 1] = static_cast<uint8_t>(std::lround(std::fmin(std::fmax((-dy * inv + 1) * 127.0f, 0.0f), 254.0f)));
            t.Pixels[4 * i + 2] = 255; t.Pixels[4 * i + 3] = 255;
        }
    return t;
}

// HDR lat-long sky (u = (1 + atan2(x, z) / pi) / 2, v = acos(y) / pi: Math::ToLatLongCoordinate): zenith gradient, clouds above
// the horizon, a ground tint and a small bright sun
inline Texture Sky(uint32_t w, uint32_t h, unsigned seed)
{
    const auto clouds = ValueNoise(w, h, seed, 4), dirt = ValueNoise(w, h, seed + 1, 3);
    Texture t; t.Width = w; t.Height = h; t.HDRPixels.resize(static_cast<size_t>(w) * h * 4);
    const float pi = 3.14159265358979323846f;
    const float sl = std::sqrt(0.4f * 0.4f + 0.6f * 0.6f + 0.7f * 0.7f), sun[3] = { 0.4f / sl, 0.6f / sl, 0.7f / sl };
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const float theta = (y + 0.5f) / h * pi, phi = (2 * (x + 0.5f) / w - 1) * pi;
            const float d[3] = { std::sin(theta) * std::sin(phi), std::cos(theta), std::sin(theta) * std::cos(phi) };
            const size_t i = static_cast<size_t>(y) * w + x;
            float c[3];
            if (d[1] >= 0) {
                const float up = std::fmin(d[1], 1.0f), cloud = std::fmax(clouds[i] - 0.5f, 0.0f) * 2 * std::fmin(d[1] * 3, 1.0f);
                c[0] = (1 - up) * 0.9f + up * 0.15f + cloud; c[1] = (1 - up) * 0.9f + up * 0.35f + cloud; c[2] = (1 - up) * 0.95f + up * 0.9f + cloud;
            } else {
                const float k = 1 + 0.3f * dirt[i];
                c[0] = 0.25f * k; c[1] = 0.22f * k; c[2] = 0.2f * k;
            }
            float s = (d[0] * sun[0] + d[1] * sun[1] + d[2] * sun[2] - 0.995f) / 0.005f;
            s = std::fmin(std::fmax(s, 0.0f), 1.0f);
            for (int k = 0; k < 3; k++) t.HDRPixels[4 * i + k] = c[k] + 40.0f * s * s;
            t.HDRPixels[4 * i + 3] = 1;
        }
    return t;
}

}  // namespace procedural

// Decoded image files (replaces the DirectXTex loaders of Source/TextureHelpers.ixx:34-138: WIC / DDS / HDR / EXR / TGA decoding is
// out of scope here -- this path takes images already decoded to 8-bit RGBA in a minimal container).  ".ptex": the ASCII tag "PTEX",
// then width and height as little-endian uint32, then width * height * 4 bytes, rows top to bottom.  tests/golden/make_textures.py
// writes the reference's own Assets/Textures in this form (decoded and reduced with PIL in the build container).
inline bool LoadRawTexture(const std::string& file, Texture& out)
{
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) return false;
    char tag[4];
    uint32_t wh[2];
    bool ok = std::fread(tag, 1, 4, f) == 4 && std::memcmp(tag, "PTEX", 4) == 0 && std::fread(wh, 4, 2, f) == 2 && wh[0] > 0 && wh[1] > 0 && wh[0] <= 16384 && wh[1] <= 16384;
    if (ok) {
        out.Width = wh[0]; out.Height = wh[1];
        out.Pixels.resize(size_t(wh[0]) * wh[1] * 4);
        ok = std::fread(out.Pixels.data(), 1, out.Pixels.size(), f) == out.Pixels.size();
    }
    std::fclose(f);
    return ok;
}

inline std::string TextureStem(const std::string& path)
{
    const size_t slash = path.find_last_of("/\\"), start = slash == std::string::npos ? 0 : slash + 1, dot = path.find_last_of('.');
    return path.substr(start, dot == std::string::npos || dot < start ? std::string::npos : dot - start);
}

// Default loader: with PT_TEXTURE_DIR set, "<dir>/<file stem>.ptex" (e.g. Earth_BaseColor.ptex for .../Earth_BaseColor.jpg), colour maps
// marked sRGB as Scene.ixx:157 does; otherwise -- and for files that directory lacks, like the reference's own missing
// Alien-Metal_Normal.png and its EXR environment -- procedural stand-ins keyed by the reference's file names (any other name gets a
// neutral grey / flat normal map).
inline Texture DefaultTextureLoader(const std::string& path, uint32_t type)
{
    if (const char* dir = std::getenv("PT_TEXTURE_DIR"); dir && *dir) {
        Texture t;
        if (LoadRawTexture(std::string(dir) + "/" + TextureStem(path) + ".ptex", t)) {
            t.ForceSRGB = type == TextureMapType::BaseColor || type == TextureMapType::EmissiveColor;
            return t;
        }
    }
    const auto has = [&](const char* s) { return path.find(s) != std::string::npos; };
    constexpr float sea[3] = { 0.05f, 0.15f, 0.45f }, land[3] = { 0.25f, 0.45f, 0.15f };
    constexpr float mare[3] = { 0.25f, 0.25f, 0.27f }, highland[3] = { 0.65f, 0.63f, 0.6f };
    constexpr float dark[3] = { 0.25f, 0.3f, 0.28f }, bright[3] = { 0.75f, 0.8f, 0.7f };
    if (type == EnvironmentLightTexture) return procedural::Sky(1024, 512, 21);  // stands in for 141_hdrmaps_com_free.exr (MyScene.ixx:95)
    if (type == TextureMapType::Normal)
        return procedural::NormalMap(has("Earth") ? 1024u : 512u, has("Earth") ? 512u : 256u, has("Earth") ? 11u : has("Moon") ? 12u : 13u, has("Moon") ? 8.0f : 4.0f);
    if (has("Earth")) return procedural::Albedo(1024, 512, 1, sea, land, 0.52f);
    if (has("Moon")) return procedural::Albedo(512, 256, 2, mare, highland, 0.45f);
    if (has("Alien-Metal") && type == TextureMapType::BaseColor) return procedural::Albedo(512, 512, 3, dark, bright, 0.5f);
    if (has("Alien-Metal") && type == TextureMapType::Metallic) return procedural::Scalar(512, 512, 4, 0.6f, 1.0f);
    if (has("Alien-Metal") && type == TextureMapType::Roughness) return procedural::Scalar(512, 512, 5, 0.15f, 0.6f);
    return procedural::Scalar(4, 4, 0, 0.5f, 0.5f);
}

}  // namespace dxrs
