// host_demo.cpp -- drives the C-ABI through the C++ host mirror exactly as INTEGRATION.md describes (the reference's
// own host is C++): MySceneDesc -> Scene::Load -> Raytracing::SetScene / SetCamera / SetConstants / Render.
// Usage: host_demo <small|demo|textured|environment> <width> <height> <bounces> <spp> <frame> <out.f32>
//   textured = the demo scene with its textured objects (MySceneDesc(seed, true)) after 3 s of motion (MyScene::SetTime)
//   environment = textured + the lat-long environment light (MySceneDesc(seed, true, true))
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "MyScene.hpp"
#include "Raytracing.hpp"

int main(int argc, char** argv)
{
    if (argc != 8) { std::fprintf(stderr, "usage: %s <small|demo|textured|environment> w h bounces spp frame out.f32\n", argv[0]); return 2; }
    try {
        const bool small = std::strcmp(argv[1], "small") == 0;
        const uint32_t w = std::atoi(argv[2]), h = std::atoi(argv[3]), bounces = std::atoi(argv[4]), spp = std::atoi(argv[5]), frame = std::atoi(argv[6]);
        dxrs::DeviceContext device;  // throws without a GPU: there is no fallback
        dxrs::Raytracing raytracing(device);
        const bool environment = std::strcmp(argv[1], "environment") == 0;
        const bool textured = environment || std::strcmp(argv[1], "textured") == 0;
        dxrs::MyScene moving(0, textured, environment);
        if (textured) moving.SetTime(3.0);  // the Earth has turned, the Moon has moved and turned with it
        dxrs::Scene still;
        if (small) still.Load(dxrs::SmallSceneDesc(0)); else still.Load(dxrs::MySceneDesc(0));
        const dxrs::Scene& scene = textured ? static_cast<const dxrs::Scene&>(moving) : still;
        const PtAccelInfo accel = raytracing.SetScene(scene);

        dxrs::CameraController controller;
        controller.SetPosition(scene.Desc.Camera.Position);                      // App::ResetCamera, Source/App.cpp:886-888
        controller.SetLens(1.57079632679489661923f, float(w) / float(h));          // HFOV 90 deg, MyAppData.h:177
        dxrs::HaltonSampler halton(8);
        dxrs::Float2 jitter{};
        for (uint32_t k = 0; k <= frame; k++) { const auto j = halton.GetNext2D(); jitter = { j.x - 0.5f, j.y - 0.5f }; }  // App.cpp:548
        dxrs::Camera camera;
        controller.Fill(camera, jitter);
        raytracing.SetCamera(camera);

        dxrs::Raytracing::GraphicsSettings gs;
        gs.RenderSize = { w, h }; gs.FrameIndex = frame; gs.Bounces = bounces; gs.SamplesPerPixel = spp; gs.IsRussianRouletteEnabled = true;
        raytracing.SetConstants(gs);
        std::vector<dxrs::Float4> radiance;
        const PtStats stats = raytracing.Render(radiance);
        std::printf("spheres %u nodes %u depth %u rays %llu ms %.3f\n", accel.leaf_count, accel.node_count, accel.depth,
                    (unsigned long long)stats.rays, stats.ms_total);
        FILE* f = std::fopen(argv[7], "wb");
        if (!f) return 3;
        std::fwrite(radiance.data(), sizeof(dxrs::Float4), radiance.size(), f);
        std::fclose(f);

        // error behaviour mirrors the reference's exceptions: an unsupported setting throws from Render
        gs.Denoiser = dxrs::Denoiser::NRDReBLUR;
        raytracing.SetConstants(gs);  // noexcept, like the reference
        bool threw = false;
        try { raytracing.Render(radiance); } catch (const std::exception& e) { threw = true; std::printf("expected error: %s\n", e.what()); }
        return threw ? 0 : 4;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
