// MyScene.hpp -- host mirror of the demo scene of Source/MyScene.ixx:52-303 at t = 0,
// plus the two benchmark scenes of SURVEY 8d that are derived from it (C1: 16 spheres, C5: 2^20 procedural).
// The environment EXR is a missing LFS blob, so by default the environment is the procedural sky (Scene.ixx:65,
// ShadingHelpers.hlsli:25-29); MySceneDesc's `environmentMap` asks for the lat-long map the reference names.
#pragma once

#include <cmath>

#include "Random.hpp"
#include "Scene.hpp"

namespace dxrs {

namespace ObjectNames {
inline constexpr const char* AlienMetal = "AlienMetal";
inline constexpr const char* Earth = "Earth";
inline constexpr const char* HarmonicOscillator = "HarmonicOscillator";
inline constexpr const char* Moon = "Moon";
inline constexpr const char* Sphere = "Sphere";
inline constexpr const char* Star = "Star";
}  // namespace ObjectNames

struct Spring { static constexpr float PositionY = 0.5f, Period = 3; };  // MyScene.ixx:49

namespace detail {

struct Hero { const char* Name; Float3 Position; dxrs::Material Material; };

inline std::vector<Hero> Heroes()  // MyScene.ixx:116-155
{
    std::vector<Hero> h(4);
    h[0].Name = ObjectNames::AlienMetal; h[0].Position = { -2, 0.5f, 0 };
    h[0].Material.BaseColor = { 1, 1, 1, 1 }; h[0].Material.Metallic = 1; h[0].Material.Roughness = 1;
    h[1].Name = ""; h[1].Position = { 0, 0.5f, 0 };
    h[1].Material.BaseColor = { 1, 1, 1, 1 }; h[1].Material.Roughness = 0; h[1].Material.Transmission = 1;
    h[2].Name = ""; h[2].Position = { 0, 2, 0 };
    h[2].Material.BaseColor = { 1, 1, 1, 1 }; h[2].Material.Roughness = 0.5f; h[2].Material.Transmission = 1;
    h[3].Name = ""; h[3].Position = { 2, 0.5f, 0 };
    h[3].Material.BaseColor = { 0.7f, 0.6f, 0.5f, 1 }; h[3].Material.Metallic = 1; h[3].Material.Roughness = 0.3f;
    return h;
}

// One of the four random material classes of MyScene.ixx:196-226 (draw order = member order).
inline Material RandomMaterial(Random& random)
{
    const auto RandomFloat4 = [&](float min) {
        const auto v = random.Float3_(min);
        return Float4{ v.x, v.y, v.z, 1 };
    };
    Material m;
    if (const auto randomValue = random.Float(); randomValue < 0.3f) {
        m.BaseColor = RandomFloat4(0.1f);
    } else if (randomValue < 0.6f) {
        m.BaseColor = RandomFloat4(0.1f);
        m.Metallic = 1;
        m.Roughness = random.Float(0, 0.5f);
    } else if (randomValue < 0.8f) {
        m.BaseColor = RandomFloat4(0.1f);
        m.Roughness = random.Float(0, 0.5f);
        m.Transmission = 1;
    } else {
        m.BaseColor = RandomFloat4(0.1f);
        m.EmissiveStrength = random.Float(1, 10);
        m.EmissiveColor = random.Float3_(0.2f);
        m.Metallic = random.Float(0.4f);
        m.Roughness = random.Float(0.3f);
    }
    return m;
}

// The 21x21 jittered grid of r = 0.075 spheres (MyScene.ixx:171-230); stops after max_count accepted.
inline void AddGrid(SceneDesc& scene, unsigned seed, size_t max_count)
{
    const auto heroes = Heroes();
    size_t accepted = 0;
    Random random(seed);
    for (int i = -10; i < 11 && accepted < max_count; i++) {
        for (int j = -10; j < 11 && accepted < max_count; j++) {
            constexpr float A = 0.5f;
            Float3 position;
            position.x = static_cast<float>(i) + 0.7f * random.Float();
            // SimpleHarmonicMotion::Spring::CalculateDisplacement(A, omega, t = 0, phi = x) = A cos(omega*0 - x)  (PhysX.h:30-31)
            position.y = Spring::PositionY + A * std::cos(0.0f - position.x);
            position.z = static_cast<float>(j) - 0.7f * random.Float();

            bool isOverlapping = false;
            for (const auto& hero : heroes) {
                const float dx = position.x - hero.Position.x, dy = position.y - hero.Position.y, dz = position.z - hero.Position.z;
                if (std::sqrt(dx * dx + dy * dy + dz * dz) < 1) { isOverlapping = true; break; }
            }
            if (isOverlapping) continue;

            RenderObjectDesc renderObject;
            renderObject.Name = ObjectNames::HarmonicOscillator;
            renderObject.Material = RandomMaterial(random);
            renderObject.Position = position;
            renderObject.Radius = 0.075f;
            scene.RenderObjects.emplace_back(renderObject);
            accepted++;
        }
    }
}

inline void AddHeroes(SceneDesc& scene)
{
    for (const auto& hero : Heroes()) {
        RenderObjectDesc o;
        o.Name = hero.Name; o.Position = hero.Position; o.Radius = 0.5f; o.Material = hero.Material;
        scene.RenderObjects.emplace_back(o);
    }
}

inline RenderObjectDesc Moon()  // MyScene.ixx:239-248
{
    RenderObjectDesc o; o.Name = ObjectNames::Moon; o.Position = { -4, 4, 0 }; o.Radius = 0.25f;
    o.Material.BaseColor = { 1, 1, 1, 1 }; o.Material.Roughness = 0.8f; return o;
}
inline RenderObjectDesc Earth()  // MyScene.ixx:249-258
{
    RenderObjectDesc o; o.Name = ObjectNames::Earth; o.Position = { 0, 4, 0 }; o.Radius = 1;
    o.Material.BaseColor = { 1, 1, 1, 1 }; o.Material.Roughness = 0.8f; return o;
}
inline RenderObjectDesc Star()  // MyScene.ixx:258-267: the mirror "ground"
{
    RenderObjectDesc o; o.Name = ObjectNames::Star; o.Position = { 0, -50.1f, 0 }; o.Radius = 50;
    o.Material.BaseColor = { 0.5f, 0.5f, 0.5f, 1 }; o.Material.Metallic = 1; o.Material.Roughness = 0; return o;
}

}  // namespace detail

// The demo default scene.  The reference seeds from random_device; the build takes a seed.
// `textured` = false is the benchmark configuration of SURVEY 8d ("textures off"); true attaches the texture files the
// reference names for Alien-Metal, Moon and Earth (MyScene.ixx:161-166, 285-295) -- resolved by the Scene's texture loader.
// `environmentMap` = true adds the reference's lat-long environment light (MyScene.ixx:94-95: yaw pi, 141_hdrmaps_com_free.exr).
struct MySceneDesc : SceneDesc {
    explicit MySceneDesc(unsigned seed = 0, bool textured = false, bool environmentMap = false)
    {
        Camera.Position.z = -15;  // MyScene.ixx:90
        if (environmentMap) {
            EnvironmentLight.Rotation = Quaternion::CreateFromYawPitchRoll(3.14159265358979323846f, 0, 0);
            EnvironmentLight.Texture = "Assets/Textures/141_hdrmaps_com_free.exr";
        }
        detail::AddHeroes(*this);
        detail::AddGrid(*this, seed, ~size_t(0));
        RenderObjects.emplace_back(detail::Moon());
        RenderObjects.emplace_back(detail::Earth());
        RenderObjects.emplace_back(detail::Star());
        if (textured) {
            const std::string dir = "Assets/Textures/";  // MyScene.ixx:92
            for (auto& o : RenderObjects) {
                if (o.Name == ObjectNames::AlienMetal) {
                    o.Textures[TextureMapType::BaseColor] = dir + "Alien-Metal_Albedo.png";
                    o.Textures[TextureMapType::Metallic] = dir + "Alien-Metal_Metallic.png";
                    o.Textures[TextureMapType::Roughness] = dir + "Alien-Metal_Roughness.png";
                    o.Textures[TextureMapType::Normal] = dir + "Alien-Metal_Normal.png";
                } else if (o.Name == ObjectNames::Moon) {
                    o.Textures[TextureMapType::BaseColor] = dir + "Moon_BaseColor.jpg";
                    o.Textures[TextureMapType::Normal] = dir + "Moon_Normal.jpg";
                } else if (o.Name == ObjectNames::Earth) {
                    o.Textures[TextureMapType::BaseColor] = dir + "Earth_BaseColor.jpg";
                    o.Textures[TextureMapType::Normal] = dir + "Earth_Normal.jpg";
                }
            }
        }
    }
};

// Scene with the demo's motion, closed form.  The reference integrates these with PhysX
// (MyScene.ixx:351-396); without collisions the default forces have exact solutions:
//   * HarmonicOscillator spheres: spring of period Spring::Period around PositionY, started at A cos(-x) with velocity
//     -A w sin(-x) (MyScene.ixx:174-178,228) -> y(t) = PositionY + A cos(w t - x), w = 2 pi / Period (PhysX.h:30-34)
//   * Moon: circular orbit of period 10 s around the Earth in the xz-plane (MyScene.ixx:240-248,270-277)
//   * Earth / Star gravity on the other bodies is off by default (userData = false), their spin only matters with textures.
struct MyScene : Scene {
    explicit MyScene(unsigned seed = 0, bool textured = false, bool environmentMap = false)
    {
        Load(MySceneDesc(seed, textured, environmentMap));
        m_initial = Desc.RenderObjects;
    }

    bool IsStatic() const { return !m_isPhysXRunning; }
    void SetRunning(bool running) { m_isPhysXRunning = running; }
    double GetTime() const { return m_time; }

    // Scene::Tick + Refresh (MyScene.ixx:310-349)
    void Tick(double elapsedSeconds)
    {
        if (IsStatic()) return;
        m_time += elapsedSeconds;
        SetTime(m_time);
    }

    void SetTime(double time)
    {
        m_time = time;
        constexpr float A = 0.5f, kTwoPi = 6.28318530717958647692f;
        const float t = static_cast<float>(time);
        for (size_t i = 0; i < Desc.RenderObjects.size(); i++) {
            auto& o = Desc.RenderObjects[i];
            const auto& o0 = m_initial[i];
            if (o.Name == ObjectNames::HarmonicOscillator) {
                const float omega = kTwoPi / Spring::Period;
                o.Position.y = Spring::PositionY + A * std::cos(omega * t - o0.Position.x);
            } else if (o.Name == ObjectNames::Moon) {
                const float theta = kTwoPi / 10.0f * t;  // OrbitalPeriod = 10 (MyScene.ixx:244)
                const float R = 4.0f;                     // |earth - moon| at t = 0
                o.Position = { -R * std::cos(theta), o0.Position.y, R * std::sin(theta) };
                o.Rotation = Quaternion::CreateFromAxisAngle({ 0, 1, 0 }, theta);  // tidally locked: angular velocity = v / R (MyScene.ixx:283)
            } else if (o.Name == ObjectNames::Earth) {
                o.Rotation = Quaternion::CreateFromAxisAngle({ 0, 1, 0 }, kTwoPi / 15.0f * t);  // RotationPeriod = 15 (MyScene.ixx:253, 289)
            }
        }
        Refresh();
    }

private:
    std::vector<RenderObjectDesc> m_initial;
    bool m_isPhysXRunning = true;
    double m_time = 0;
};

// 16 spheres = 4 heroes + first 10 accepted grid spheres (seed) + Earth-like + Star ground.
struct SmallSceneDesc : SceneDesc {
    explicit SmallSceneDesc(unsigned seed = 0)
    {
        Camera.Position.z = -15;
        detail::AddHeroes(*this);
        detail::AddGrid(*this, seed, 10);
        RenderObjects.emplace_back(detail::Earth());
        RenderObjects.emplace_back(detail::Star());
    }
};

// `count` spheres, centres uniform in x,z in [-200,200], y in [0.1,20], radius log-uniform
// in [0.02,0.2], material classes as the grid's 30/30/20/20 split, plus the Star ground as the last object.
struct ProceduralSceneDesc : SceneDesc {
    explicit ProceduralSceneDesc(uint32_t count, unsigned seed = 1)
    {
        Camera.Position.z = -15;
        Random random(seed);
        RenderObjects.reserve(static_cast<size_t>(count) + 1);
        for (uint32_t i = 0; i < count; i++) {
            RenderObjectDesc o;
            o.Name = ObjectNames::Sphere;
            o.Position.x = random.Float(-200, 200);
            o.Position.y = random.Float(0.1f, 20);
            o.Position.z = random.Float(-200, 200);
            o.Radius = 0.02f * std::pow(10.0f, random.Float());
            o.Material = detail::RandomMaterial(random);
            RenderObjects.emplace_back(o);
        }
        RenderObjects.emplace_back(detail::Star());
    }
};

}  // namespace dxrs
