// Scene.hpp -- host mirror of the scene-side producers of Source/Scene.ixx: RenderObjectDesc / SceneDesc
// (:33-85) and Scene::{Load, Refresh, GetObjectCount} (:123-219).  PhysX rigid bodies are replaced by a
// plain pose (position + radius): Refresh() applies the reference's world transform
// diag(1,1,-1) * pose * scale(2r) (:188-203) to the unit-diameter sphere, i.e. centre = (x, y, -z), radius = r.
#pragma once

#include <string>
#include <vector>

#include <array>
#include <map>

#include "Camera.hpp"
#include "Material.hpp"
#include "Texture.hpp"

namespace dxrs {

struct RenderObjectDesc {
    std::string Name;
    Float3 Position;       // PhysX-space centre (z is flipped on Refresh)
    float Radius = 0.5f;   // PxSphereGeometry radius
    dxrs::Material Material;
    Quaternion Rotation;   // PhysX-space pose rotation (mirrored with the position on Refresh)
    std::array<std::string, TextureMapType::Count> Textures;  // file per TextureMapType, empty = none (Scene.ixx:40)
};

struct SceneDesc {
    struct {
        Float3 Position;
        Quaternion Rotation;
    } Camera;

    struct {
        Float4 Color{ 0, 0, 0, -1 };  // Scene.ixx:65: a < 0 -> procedural sky
        Quaternion Rotation;
        std::string Texture;          // Scene.ixx:78: a lat-long environment map file, empty = none
    } EnvironmentLight;

    std::vector<RenderObjectDesc> RenderObjects;
};

struct Scene {
    SceneDesc Desc;

    void SetTextureLoader(TextureLoader loader) { m_loader = std::move(loader); }

    // Scene::Load (Scene.ixx:123-180): every distinct texture file is loaded once and shared by the objects that name it
    void Load(const SceneDesc& sceneDesc)
    {
        Desc = sceneDesc;
        m_textures.clear();
        m_objectTextures.assign(Desc.RenderObjects.size(), PtObjectTextures{});
        std::map<std::pair<std::string, uint32_t>, uint32_t> loaded;
        for (size_t i = 0; i < Desc.RenderObjects.size(); i++)
            for (uint32_t k = 0; k < TextureMapType::Count; k++) {
                auto& info = m_objectTextures[i].Maps[k];
                info.Descriptor = ~0u;
                const auto& path = Desc.RenderObjects[i].Textures[k];
                if (path.empty()) continue;
                const auto key = std::make_pair(path, k);
                auto it = loaded.find(key);
                if (it == loaded.end()) {
                    m_textures.emplace_back(m_loader ? m_loader(path, k) : DefaultTextureLoader(path, k));
                    it = loaded.emplace(key, static_cast<uint32_t>(m_textures.size() - 1)).first;
                }
                info.Descriptor = it->second;
            }
        m_environmentDescriptor = ~0u;
        m_environmentIsCubeMap = false;
        if (!Desc.EnvironmentLight.Texture.empty()) {  // Scene.ixx:130-133
            auto texture = m_loader ? m_loader(Desc.EnvironmentLight.Texture, EnvironmentLightTexture) : DefaultTextureLoader(Desc.EnvironmentLight.Texture, EnvironmentLightTexture);
            m_environmentDescriptor = static_cast<uint32_t>(m_textures.size());
            m_environmentIsCubeMap = texture.IsCubeMap();
            if (m_environmentIsCubeMap) for (auto& face : texture.Faces) m_textures.emplace_back(std::move(face));  // six consecutive table entries
            else m_textures.emplace_back(std::move(texture));
        }
        Refresh();
    }

    // Scene.ixx:185-219
    void Refresh()
    {
        const auto n = Desc.RenderObjects.size();
        m_spheres.resize(n);
        m_materials.resize(n);
        m_rotations.resize(4 * n);
        for (size_t i = 0; i < n; i++) {
            const auto& o = Desc.RenderObjects[i];
            m_spheres[i] = PtSphere{ o.Position.x, o.Position.y, -o.Position.z, o.Radius };
            m_materials[i] = ToPt(o.Material);
            // the z mirror diag(1,1,-1) conjugates a rotation (x, y, z, w) into (-x, -y, z, w)
            m_rotations[4 * i] = -o.Rotation.x; m_rotations[4 * i + 1] = -o.Rotation.y;
            m_rotations[4 * i + 2] = o.Rotation.z; m_rotations[4 * i + 3] = o.Rotation.w;
        }
    }

    uint32_t GetObjectCount() const noexcept { return static_cast<uint32_t>(m_spheres.size()); }
    const std::vector<PtSphere>& GetSpheres() const noexcept { return m_spheres; }
    const std::vector<PtMaterial>& GetMaterials() const noexcept { return m_materials; }
    bool HasTextures() const noexcept { return !m_textures.empty(); }
    const std::vector<Texture>& GetTextures() const noexcept { return m_textures; }
    const std::vector<PtObjectTextures>& GetObjectTextures() const noexcept { return m_objectTextures; }
    const std::vector<float>& GetRotations() const noexcept { return m_rotations; }  // n x (x, y, z, w), world space

    // SceneData as uploaded by App::UpdateScene (Source/App.cpp:977-990).  EnvironmentLightTransform =
    // XMStoreFloat3x4(Matrix::CreateFromQuaternion(Rotation)): the 3x4 store transposes DirectXMath's row-vector matrix, so
    // the shader's mul((float3x3)M, v) rotates v by the quaternion.
    PtSceneData GetSceneData() const
    {
        PtSceneData sd{};
        sd.IsStatic = 1;
        sd.EnvironmentLightTextureDescriptor = m_environmentDescriptor;
        sd.IsEnvironmentLightTextureCubeMap = m_environmentIsCubeMap ? 1u : 0u;
        sd.EnvironmentLightColor[0] = Desc.EnvironmentLight.Color.x;
        sd.EnvironmentLightColor[1] = Desc.EnvironmentLight.Color.y;
        sd.EnvironmentLightColor[2] = Desc.EnvironmentLight.Color.z;
        sd.EnvironmentLightColor[3] = Desc.EnvironmentLight.Color.w;
        const auto& q = Desc.EnvironmentLight.Rotation;
        float* m = sd.EnvironmentLightTransform;
        m[0] = 1 - 2 * (q.y * q.y + q.z * q.z); m[1] = 2 * (q.x * q.y - q.z * q.w); m[2] = 2 * (q.x * q.z + q.y * q.w);
        m[4] = 2 * (q.x * q.y + q.z * q.w); m[5] = 1 - 2 * (q.x * q.x + q.z * q.z); m[6] = 2 * (q.y * q.z - q.x * q.w);
        m[8] = 2 * (q.x * q.z - q.y * q.w); m[9] = 2 * (q.y * q.z + q.x * q.w); m[10] = 1 - 2 * (q.x * q.x + q.y * q.y);
        return sd;
    }

private:
    std::vector<PtSphere> m_spheres;
    std::vector<PtMaterial> m_materials;
    std::vector<float> m_rotations;
    std::vector<Texture> m_textures;
    std::vector<PtObjectTextures> m_objectTextures;
    uint32_t m_environmentDescriptor = ~0u;
    bool m_environmentIsCubeMap = false;
    TextureLoader m_loader;
};

}  // namespace dxrs
