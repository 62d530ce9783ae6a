// HaltonSampler.hpp -- host mirror of Source/HaltonSampler.ixx:19-45: cyclic Halton(2,3,5) sampler,
// index starts at 1 and wraps at `count`.  Sequence::Halton1D/2D/3D come from the un-vendored
// MathLib; restated per SURVEY Appendix A (base 2 = bit reversal * 2^-32, other bases the float loop).
#pragma once

#include <cstdint>
#include <stdexcept>

#include "Material.hpp"

namespace dxrs {

inline float Halton(uint32_t index, uint32_t base)
{
    if (base == 2) {
        uint32_t v = index;
        v = (v << 16) | (v >> 16);
        v = ((v & 0x00FF00FFu) << 8) | ((v & 0xFF00FF00u) >> 8);
        v = ((v & 0x0F0F0F0Fu) << 4) | ((v & 0xF0F0F0F0u) >> 4);
        v = ((v & 0x33333333u) << 2) | ((v & 0xCCCCCCCCu) >> 2);
        v = ((v & 0x55555555u) << 1) | ((v & 0xAAAAAAAAu) >> 1);
        return static_cast<float>(v) * 2.3283064365386963e-10f;
    }
    float f = 1.0f, r = 0.0f;
    const float fb = static_cast<float>(base);
    for (uint32_t i = index; i > 0; i /= base) {
        f = f / fb;
        r = r + f * static_cast<float>(i % base);
    }
    return r;
}

class HaltonSampler {
public:
    explicit HaltonSampler(uint32_t count = ~0u) noexcept(false) : m_count(count)
    {
        if (!count) throw std::out_of_range("Sample count cannot be 0");
    }

    static float Get1D(uint32_t index) { return Halton(index, 2); }
    static Float2 Get2D(uint32_t index) { return { Halton(index, 2), Halton(index, 3) }; }
    static Float3 Get3D(uint32_t index) { return { Halton(index, 2), Halton(index, 3), Halton(index, 5) }; }

    float GetNext1D() noexcept { auto r = Get1D(m_index + 1); Advance(); return r; }
    Float2 GetNext2D() noexcept { auto r = Get2D(m_index + 1); Advance(); return r; }
    Float3 GetNext3D() noexcept { auto r = Get3D(m_index + 1); Advance(); return r; }

    uint32_t GetCount() const noexcept { return m_count; }
    uint32_t GetIndex() const noexcept { return m_index; }
    void Reset() noexcept { m_index = 0; }

private:
    void Advance() noexcept { m_index = (m_index + 1) % m_count; }

    uint32_t m_count, m_index{};
};

}  // namespace dxrs
