// Material.hpp -- host mirror of the reference's Material API struct (Source/Material.ixx:10-38).
// Same field names, defaults and 64-byte layout, so host code written against the reference's
// `Material` compiles against this one; it is bit-compatible with PtMaterial of the C-ABI.
#pragma once

#include <cstdint>
#include <cstring>

#include "../../include/pt_types.h"

namespace dxrs {

struct Float2 { float x{}, y{}; };
struct Float3 { float x{}, y{}, z{}; };
struct Float4 { float x{}, y{}, z{}, w{}; };
struct UInt2 { uint32_t x{}, y{}; };

enum class AlphaMode : uint32_t { Opaque, Mask, Blend };  // Material.ixx:10

struct Material {  // Material.ixx:12-20
    Float4 BaseColor{ 0, 0, 0, 1 };
    float EmissiveStrength = 1;
    Float3 EmissiveColor{};
    float Metallic{}, Roughness = 0.5f, IOR = 1.5f, Transmission{};
    dxrs::AlphaMode AlphaMode = dxrs::AlphaMode::Opaque;
    float AlphaCutoff = 0.5f;
    UInt2 _{};
};
static_assert(sizeof(Material) == sizeof(PtMaterial), "Material must match the C-ABI layout");

struct TextureMapType {  // Material.ixx:22-33
    enum : uint32_t { BaseColor, EmissiveColor, Metallic, Roughness, MetallicRoughness, Transmission, Normal, Count };
};

struct TextureMapInfo {  // Material.ixx:35-38
    uint32_t Descriptor = ~0u, TextureCoordinateIndex{};
    UInt2 _{};
};

inline PtMaterial ToPt(const Material& m)
{
    PtMaterial r;
    std::memcpy(&r, &m, sizeof r);
    return r;
}

}  // namespace dxrs
