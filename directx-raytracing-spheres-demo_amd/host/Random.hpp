// Random.hpp -- host mirror of Source/Random.ixx:12-32 (scene generation only; NOT the per-pixel RNG).
// The reference seeds mt19937 from random_device and maps through uniform_real_distribution<float>
// (implementation-defined).  The build makes it deterministic: explicit seed, and
// u = float(raw32) * 2^-32 clamped below 1 ("Random -> float mapping").
#pragma once

#include <random>

#include "Material.hpp"

namespace dxrs {

struct Random {
    explicit Random(unsigned int seed = 0) : m_generator(seed) {}

    float Unit()
    {
        float u = static_cast<float>(m_generator()) * 2.3283064365386963e-10f;
        return u >= 1.0f ? 0.99999994f : u;
    }

    float Float(float min = 0, float max = 1) { return min + (max - min) * Unit(); }

    Float2 Float2_(float min = 0, float max = 1)
    {
        const float x = Float(min, max);
        const float y = Float(min, max);
        return { x, y };
    }

    Float3 Float3_(float min = 0, float max = 1)
    {
        const auto value = Float2_(min, max);
        const float z = Float(min, max);
        return { value.x, value.y, z };
    }

    Float4 Float4_(float min = 0, float max = 1)
    {
        const auto value = Float3_(min, max);
        const float w = Float(min, max);
        return { value.x, value.y, value.z, w };
    }

private:
    std::mt19937 m_generator;
};

}  // namespace dxrs
