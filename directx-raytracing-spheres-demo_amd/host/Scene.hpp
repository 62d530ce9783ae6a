// Scene.hpp -- host mirror of the scene-side producers of Source/Scene.ixx: RenderObjectDesc / SceneDesc
// (:33-85) and Scene::{Load, Refresh, GetObjectCount} (:123-219).  PhysX rigid bodies are replaced by a
// plain pose (position + radius): Refresh() applies the reference's world transform
// diag(1,1,-1) * pose * scale(2r) (:188-203) to the unit-diameter sphere, i.e. centre = (x, y, -z), radius = r.
#pragma once

#include <string>
#include <vector>

#include "Camera.hpp"
#include "Material.hpp"

namespace dxrs {

struct RenderObjectDesc {
    std::string Name;
    Float3 Position;       // PhysX-space centre (z is flipped on Refresh)
    float Radius = 0.5f;   // PxSphereGeometry radius
    dxrs::Material Material;
};

struct SceneDesc {
    struct {
        Float3 Position;
        Quaternion Rotation;
    } Camera;

    struct {
        Float4 Color{ 0, 0, 0, -1 };  // Scene.ixx:65: a < 0 -> procedural sky
        Quaternion Rotation;
    } EnvironmentLight;

    std::vector<RenderObjectDesc> RenderObjects;
};

struct Scene {
    SceneDesc Desc;

    void Load(const SceneDesc& sceneDesc)
    {
        Desc = sceneDesc;
        Refresh();
    }

    // Scene.ixx:185-219
    void Refresh()
    {
        const auto n = Desc.RenderObjects.size();
        m_spheres.resize(n);
        m_materials.resize(n);
        for (size_t i = 0; i < n; i++) {
            const auto& o = Desc.RenderObjects[i];
            m_spheres[i] = PtSphere{ o.Position.x, o.Position.y, -o.Position.z, o.Radius };
            m_materials[i] = ToPt(o.Material);
        }
    }

    uint32_t GetObjectCount() const noexcept { return static_cast<uint32_t>(m_spheres.size()); }
    const std::vector<PtSphere>& GetSpheres() const noexcept { return m_spheres; }
    const std::vector<PtMaterial>& GetMaterials() const noexcept { return m_materials; }

    // SceneData as uploaded by App::UpdateScene (Source/App.cpp:977-990) with no environment texture.
    PtSceneData GetSceneData() const
    {
        PtSceneData sd{};
        sd.IsStatic = 1;
        sd.EnvironmentLightTextureDescriptor = ~0u;
        sd.EnvironmentLightColor[0] = Desc.EnvironmentLight.Color.x;
        sd.EnvironmentLightColor[1] = Desc.EnvironmentLight.Color.y;
        sd.EnvironmentLightColor[2] = Desc.EnvironmentLight.Color.z;
        sd.EnvironmentLightColor[3] = Desc.EnvironmentLight.Color.w;
        sd.EnvironmentLightTransform[0] = sd.EnvironmentLightTransform[5] = sd.EnvironmentLightTransform[10] = 1;
        return sd;
    }

private:
    std::vector<PtSphere> m_spheres;
    std::vector<PtMaterial> m_materials;
};

}  // namespace dxrs
