// Raytracing.hpp -- host mirror of the reference's pass objects for this path, on top of the C-ABI:
//   Raytracing        Source/Raytracing.ixx:29-112  { GraphicsSettings, SetConstants(...) noexcept, Render(...) }
//   (GBufferGeneration Source/GBufferGeneration.ixx:27-117 is folded in: Render traces the primary hit too.)
// Same names, argument meaning and error behaviour: errors surface as C++ exceptions
// (reference: ThrowIfFailed -> std::system_error, Source/ErrorHelpers.ixx:16-32); SetConstants is noexcept.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "Camera.hpp"
#include "HaltonSampler.hpp"
#include "Scene.hpp"

namespace dxrs {

enum class Denoiser : uint32_t { None, DLSSRayReconstruction, NRDReBLUR, NRDReLAX };  // Source/Denoiser.ixx

inline void ThrowIfFailed(PtStatus status, PtContext* ctx, const char* what)
{
    if (status != PT_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(static_cast<int>(status)) + "): " + pt_last_error(ctx));
}

// Stand-in for the reference's DeviceContext/CommandList pair: owns the PtContext (device, stream, device memory).
class DeviceContext {
public:
    explicit DeviceContext(int device = 0, uint32_t flags = 0, uint64_t stream = 0, uint32_t tileSize = 0)
    {
        PtConfig config{};
        config.device = device; config.flags = flags; config.stream = stream; config.tile_size = tileSize;
        PtContext* ctx = nullptr;
        const PtStatus st = pt_create(&config, &ctx);
        if (st != PT_OK) throw std::runtime_error("pt_create failed (" + std::to_string(static_cast<int>(st)) + "): no usable HIP device");
        m_ctx.reset(ctx);
    }
    PtContext* Get() const noexcept { return m_ctx.get(); }

private:
    struct Deleter { void operator()(PtContext* p) const noexcept { pt_destroy(p); } };
    std::unique_ptr<PtContext, Deleter> m_ctx;
};

struct Raytracing {
    struct GraphicsSettings {  // Raytracing.ixx:30-36
        UInt2 RenderSize;
        uint32_t FrameIndex{}, Bounces{}, SamplesPerPixel{};
        float ThroughputThreshold = 1e-3f;
        bool IsRussianRouletteEnabled{}, IsShaderExecutionReorderingEnabled{}, IsDIEnabled{};
        dxrs::Denoiser Denoiser = dxrs::Denoiser::None;
    };

    explicit Raytracing(DeviceContext& deviceContext) noexcept(false) : m_ctx(deviceContext.Get())
    {
        if (!m_ctx) throw std::invalid_argument("null device context");
    }

    // Scene::Load/Refresh + CreateAccelerationStructures (Scene.ixx:123-284)
    PtAccelInfo SetScene(const Scene& scene)
    {
        const auto sd = scene.GetSceneData();
        ThrowIfFailed(pt_set_scene(m_ctx, scene.GetSpheres().data(), scene.GetMaterials().data(), scene.GetObjectCount(), &sd), m_ctx, "pt_set_scene");
        PtAccelInfo info{};
        ThrowIfFailed(pt_build_accel(m_ctx, &info), m_ctx, "pt_build_accel");
        if (scene.HasTextures()) {  // ObjectData::TextureMapInfoArray + the texture uploads of Scene::Load (Scene.ixx:150-180)
            std::vector<PtTexture> textures;
            for (const auto& t : scene.GetTextures()) textures.push_back(t.ToPt());
            ThrowIfFailed(pt_set_textures(m_ctx, textures.data(), static_cast<uint32_t>(textures.size()), scene.GetObjectTextures().data(),
                                          scene.GetRotations().data()), m_ctx, "pt_set_textures");
        }
        return info;
    }

    // the poses of a running scene (MyScene::Tick -> Scene::Refresh): rotations only orient texture coordinates
    void UpdateRotations(const Scene& scene)
    {
        if (scene.HasTextures())
            ThrowIfFailed(pt_update_rotations(m_ctx, scene.GetRotations().data(), scene.GetObjectCount()), m_ctx, "pt_update_rotations");
    }

    void SetCamera(const Camera& camera)
    {
        const auto cam = ToPt(camera);
        ThrowIfFailed(pt_set_camera(m_ctx, &cam), m_ctx, "pt_set_camera");
    }

    void SetConstants(const GraphicsSettings& graphicsSettings) noexcept  // Raytracing.ixx:92-104
    {
        m_graphicsSettings = PtGraphicsSettings{};
        m_graphicsSettings.RenderSize[0] = graphicsSettings.RenderSize.x;
        m_graphicsSettings.RenderSize[1] = graphicsSettings.RenderSize.y;
        m_graphicsSettings.FrameIndex = graphicsSettings.FrameIndex;
        m_graphicsSettings.Bounces = graphicsSettings.Bounces;
        m_graphicsSettings.SamplesPerPixel = graphicsSettings.SamplesPerPixel;
        m_graphicsSettings.ThroughputThreshold = graphicsSettings.ThroughputThreshold;
        m_graphicsSettings.IsRussianRouletteEnabled = graphicsSettings.IsRussianRouletteEnabled;
        m_graphicsSettings.IsShaderExecutionReorderingEnabled = graphicsSettings.IsShaderExecutionReorderingEnabled;
        m_graphicsSettings.IsDIEnabled = graphicsSettings.IsDIEnabled;
        m_graphicsSettings.Denoiser = static_cast<uint32_t>(graphicsSettings.Denoiser);
    }

    // the constants alone (what Render does first): for callers that dispatch through the tile entry points (TileExchange.hpp)
    void UploadConstants() { ThrowIfFailed(pt_set_constants(m_ctx, &m_graphicsSettings), m_ctx, "pt_set_constants"); }

    // Raytracing::Render (Raytracing.ixx:106-112): uploads the constants, then DispatchRays(W, H, 1).
    // radiance: W*H float4 host buffer (the reference's Radiance texture, kept at fp32).
    PtStats Render(std::vector<Float4>& radiance)
    {
        ThrowIfFailed(pt_set_constants(m_ctx, &m_graphicsSettings), m_ctx, "pt_set_constants");
        radiance.resize(static_cast<size_t>(m_graphicsSettings.RenderSize[0]) * m_graphicsSettings.RenderSize[1]);
        PtStats stats{};
        ThrowIfFailed(pt_render(m_ctx, nullptr, radiance.data(), 0, &stats), m_ctx, "pt_render");
        return stats;
    }

private:
    PtContext* m_ctx;
    PtGraphicsSettings m_graphicsSettings{};
};

}  // namespace dxrs
