// Camera.hpp -- host mirror of Source/Camera.ixx: the `Camera` constant-buffer struct (:16-36, same
// field names and byte layout as PtCamera) and `CameraController` (:38-177).  The bounce loop reads only
// Position / Right / Up / Forward (lens-scaled, un-normalised), NearDepth, FarDepth and Jitter
// (Shaders/Camera.hlsli:27-41); the eight matrices belong to dropped passes (motion vectors, denoisers)
// and are left zero.
#pragma once

#include <cmath>
#include <cstring>
#include <limits>

#include "Material.hpp"

namespace dxrs {

struct alignas(256) Camera {  // Camera.ixx:16-36 (D3D12_CONSTANT_BUFFER_DATA_PLACEMENT_ALIGNMENT = 256)
    uint32_t IsNormalizedDepthReversed{};
    Float3 PreviousPosition, Position;
    float _{};
    Float3 RightDirection;
    float _1{};
    Float3 UpDirection;
    float _2{};
    Float3 ForwardDirection;
    float ApertureRadius{}, NearDepth{}, FarDepth{};
    Float2 Jitter;
    float Matrices[8][16]{};
};
static_assert(sizeof(Camera) == 768, "alignas(256) payload of 608 B");

inline PtCamera ToPt(const Camera& c)
{
    PtCamera r;
    std::memcpy(&r, &c, sizeof r);
    return r;
}

inline Float3 operator+(Float3 a, Float3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Float3 operator-(Float3 a, Float3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Float3 operator*(Float3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Float3 operator/(Float3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }

namespace detail {
inline float Length(Float3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Float3 Cross(Float3 a, Float3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline Float3 Normalized(Float3 a) { return a / Length(a); }
}  // namespace detail

struct Quaternion {
    float x{}, y{}, z{}, w = 1;

    static Quaternion CreateFromAxisAngle(Float3 axis, float angle)
    {
        const auto n = detail::Normalized(axis);
        const float s = std::sin(angle * 0.5f);
        return { n.x * s, n.y * s, n.z * s, std::cos(angle * 0.5f) };
    }
    static Quaternion CreateFromYawPitchRoll(float yaw, float pitch, float roll)
    {
        const float cy = std::cos(yaw * 0.5f), sy = std::sin(yaw * 0.5f);
        const float cp = std::cos(pitch * 0.5f), sp = std::sin(pitch * 0.5f);
        const float cr = std::cos(roll * 0.5f), sr = std::sin(roll * 0.5f);
        return { cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr, cy * cp * sr - sy * sp * cr, cy * cp * cr + sy * sp * sr };
    }
    Quaternion Normalized() const
    {
        const float l = std::sqrt(x * x + y * y + z * z + w * w);
        return { x / l, y / l, z / l, w / l };
    }
    // SimpleMath semantics: (a * b) applies a first, then b.
    friend Quaternion operator*(const Quaternion& a, const Quaternion& b)
    {
        return { b.w * a.x + b.x * a.w + b.y * a.z - b.z * a.y,
                 b.w * a.y - b.x * a.z + b.y * a.w + b.z * a.x,
                 b.w * a.z + b.x * a.y - b.y * a.x + b.z * a.w,
                 b.w * a.w - b.x * a.x - b.y * a.y - b.z * a.z };
    }
    Float3 Rotate(Float3 v) const
    {
        using namespace detail;
        const Float3 u{ x, y, z };
        const Float3 t = Cross(u, v) * 2.0f;
        return v + t * w + Cross(u, t);
    }
};

struct CameraController {  // Camera.ixx:38-177
    explicit CameraController(bool isNormalizedDepthReversed = true) : m_isNormalizedDepthReversed(isNormalizedDepthReversed) {}

    const Float3& GetPosition() const { return m_position; }
    void SetPosition(const Float3& value) { m_position = value; }

    const Float3& GetRightDirection() const { return m_rightDirection; }
    const Float3& GetUpDirection() const { return m_upDirection; }
    const Float3& GetForwardDirection() const { return m_forwardDirection; }

    Float3 GetNormalizedRightDirection() const { return detail::Normalized(m_rightDirection); }
    Float3 GetNormalizedUpDirection() const { return detail::Normalized(m_upDirection); }
    Float3 GetNormalizedForwardDirection() const { return detail::Normalized(m_forwardDirection); }

    void SetDirections(const Float3& forwardDirection, const Float3& upDirection = { 0, 1, 0 }, bool setFocusDistance = true)
    {
        using namespace detail;
        m_forwardDirection = forwardDirection;
        m_rightDirection = Cross(upDirection, forwardDirection);
        m_upDirection = Cross(m_forwardDirection, m_rightDirection);
        if (setFocusDistance) {
            SetFocusDistance(Length(m_forwardDirection));
        } else {
            m_rightDirection = GetNormalizedRightDirection() * m_rightDirectionLength;
            m_upDirection = GetNormalizedUpDirection() * m_upDirectionLength;
            m_forwardDirection = GetNormalizedForwardDirection() * m_forwardDirectionLength;
        }
    }

    const Quaternion& GetRotation() const { return m_rotation; }

    void SetRotation(const Quaternion& value)
    {
        using namespace detail;
        m_rotation = value.Normalized();
        m_forwardDirection = Normalized(m_rotation.Rotate({ 0, 0, 1 }));
        m_rightDirection = Normalized(m_rotation.Rotate({ 1, 0, 0 }));
        m_upDirection = Cross(m_forwardDirection, m_rightDirection) * m_upDirectionLength;
        m_forwardDirection = m_forwardDirection * m_forwardDirectionLength;
        m_rightDirection = m_rightDirection * m_rightDirectionLength;
    }

    void LookAt(const Float3& position, const Float3& upDirection = { 0, 1, 0 }, bool setFocusDistance = true)
    {
        using namespace detail;
        SetDirections(position - m_position, upDirection, setFocusDistance);
    }

    float GetFocusDistance() const { return m_forwardDirectionLength; }

    void SetFocusDistance(float value)
    {
        m_rightDirectionLength *= value / m_forwardDirectionLength;
        m_upDirectionLength *= value / m_forwardDirectionLength;
        m_forwardDirectionLength = value;
        m_rightDirection = GetNormalizedRightDirection() * m_rightDirectionLength;
        m_upDirection = GetNormalizedUpDirection() * m_upDirectionLength;
        m_forwardDirection = GetNormalizedForwardDirection() * m_forwardDirectionLength;
    }

    void Translate(const Float3& value) { using namespace detail; SetPosition(m_position + value); }

    void Rotate(float yaw, float pitch, float roll = 0)
    {
        SetRotation(m_rotation * Quaternion::CreateFromAxisAngle(m_rightDirection, -pitch) * Quaternion::CreateFromAxisAngle({ 0, 1, 0 }, yaw)
                    * Quaternion::CreateFromAxisAngle(m_forwardDirection, -roll));
    }

    float GetHorizontalFieldOfView() const { return m_horizontalFieldOfView; }
    float GetVerticalFieldOfView() const { return 2 * std::atan(std::tan(m_horizontalFieldOfView / 2) * m_aspectRatio); }
    float GetAspectRatio() const { return m_aspectRatio; }
    float GetNearDepth() const { return m_nearDepth; }
    float GetFarDepth() const { return m_farDepth; }

    void SetLens(float horizontalFieldOfView, float aspectRatio)
    {
        using namespace detail;
        m_horizontalFieldOfView = horizontalFieldOfView;
        m_aspectRatio = aspectRatio;
        m_rightDirectionLength = std::tan(horizontalFieldOfView / 2) * m_forwardDirectionLength;
        m_upDirectionLength = m_rightDirectionLength / aspectRatio;
        m_upDirection = GetNormalizedUpDirection() * m_upDirectionLength;
        m_rightDirection = GetNormalizedRightDirection() * m_rightDirectionLength;
    }

    void SetLens(float horizontalFieldOfView, float aspectRatio, float nearDepth, float farDepth = std::numeric_limits<float>::infinity())
    {
        SetLens(horizontalFieldOfView, aspectRatio);
        m_nearDepth = nearDepth;
        m_farDepth = farDepth;
    }

    // The camera block of App::Impl::Update (Source/App.cpp:531-553), minus the matrices.
    void Fill(Camera& camera, Float2 jitter) const
    {
        camera.IsNormalizedDepthReversed = m_isNormalizedDepthReversed;
        camera.PreviousPosition = camera.Position;
        camera.Position = m_position;
        camera.RightDirection = m_rightDirection;
        camera.UpDirection = m_upDirection;
        camera.ForwardDirection = m_forwardDirection;
        camera.NearDepth = m_nearDepth;
        camera.FarDepth = m_farDepth;
        camera.Jitter = jitter;
    }

private:
    bool m_isNormalizedDepthReversed;
    float m_rightDirectionLength = 1, m_upDirectionLength = 1, m_forwardDirectionLength = 1;
    Float3 m_position, m_rightDirection{ 1, 0, 0 }, m_upDirection{ 0, 1, 0 }, m_forwardDirection{ 0, 0, 1 };
    Quaternion m_rotation;
    float m_horizontalFieldOfView{}, m_aspectRatio{}, m_nearDepth = 1e-2f, m_farDepth = std::numeric_limits<float>::infinity();
};

}  // namespace dxrs
