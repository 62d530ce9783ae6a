// host_tiles.cpp -- a C++ host rendering a frame as tiles on N GPUs through the C-ABI alone (no Python, no HIP headers):
// DeviceContext -> Raytracing::SetScene -> pt_comm_init -> TileExchange<PtBackend>{Submit, Finish} -> pt_download.
// One process per GPU: RANK / WORLD_SIZE / LOCAL_RANK from the environment (default: one rank); the RCCL id travels through a file
// (rank 0 writes it, the others wait for it) -- any launcher-side channel would do.  Rank 0 writes the last frame as raw float4.
// Usage: host_tiles <w> <h> <bounces> <spp> <frames> <batch> <root_weight> <out.f32> [id_file]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "MyScene.hpp"
#include "TileExchange.hpp"

static uint32_t EnvU32(const char* name, uint32_t dflt) { const char* v = std::getenv(name); return v && *v ? uint32_t(std::atoi(v)) : dflt; }

int main(int argc, char** argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s w h bounces spp frames batch root_weight out.f32 [id_file]\n", argv[0]); return 2; }
    try {
        const uint32_t w = std::atoi(argv[1]), h = std::atoi(argv[2]), bounces = std::atoi(argv[3]), spp = std::atoi(argv[4]);
        const uint32_t frames = std::atoi(argv[5]), batch = std::atoi(argv[6]), weight = std::atoi(argv[7]);
        const uint32_t rank = EnvU32("RANK", 0), world = EnvU32("WORLD_SIZE", 1), local = EnvU32("LOCAL_RANK", rank);
        dxrs::DeviceContext device(int(local), 0, 0, 32);
        PtContext* ctx = device.Get();
        dxrs::Raytracing raytracing(device);
        dxrs::Scene scene;
        scene.Load(dxrs::MySceneDesc(0));
        raytracing.SetScene(scene);  // the scene + BVH are replicated on every GPU

        // communicator (also with one rank: loads RCCL and creates a one-rank communicator -- the call path of the N-rank job)
        unsigned char id[PT_COMM_ID_BYTES];
        const std::string idFile = argc > 9 ? argv[9] : "";
        if (rank == 0) {
            dxrs::ThrowIfFailed(pt_comm_unique_id(id), ctx, "pt_comm_unique_id");
            if (world > 1) {
                FILE* f = std::fopen((idFile + ".tmp").c_str(), "wb");
                if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) throw std::runtime_error("cannot write the id file");
                std::fclose(f);
                std::rename((idFile + ".tmp").c_str(), idFile.c_str());
            }
        } else {
            for (int tries = 0;; tries++) {
                FILE* f = std::fopen(idFile.c_str(), "rb");
                if (f) { const size_t n = std::fread(id, 1, sizeof id, f); std::fclose(f); if (n == sizeof id) break; }
                if (tries > 600) throw std::runtime_error("timed out waiting for the id file");
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        dxrs::ThrowIfFailed(pt_comm_init(ctx, id, rank, world), ctx, "pt_comm_init");

        dxrs::CameraController controller;
        controller.SetPosition(scene.Desc.Camera.Position);
        controller.SetLens(1.57079632679489661923f, float(w) / float(h));
        dxrs::Raytracing::GraphicsSettings gs;
        gs.RenderSize = { w, h }; gs.Bounces = bounces; gs.SamplesPerPixel = spp; gs.IsRussianRouletteEnabled = true;
        std::vector<dxrs::Float2> jitters;
        { dxrs::HaltonSampler halton(8); for (uint32_t k = 0; k < 8; k++) { const auto j = halton.GetNext2D(); jitters.push_back({ j.x - 0.5f, j.y - 0.5f }); } }
        auto setFrame = [&](uint32_t k) {
            dxrs::Camera camera;
            controller.Fill(camera, jitters[k % 8]);
            raytracing.SetCamera(camera);
            gs.FrameIndex = k;
            raytracing.SetConstants(gs);
            raytracing.UploadConstants();
        };
        setFrame(0);  // pt_render_tiles / the tile counts need RenderSize
        dxrs::PtBackend backend(ctx, setFrame);
        dxrs::TileExchange<dxrs::PtBackend> exchange(backend, w, h, rank, world, batch, true, 32);
        exchange.Configure(weight);
        for (uint32_t k = 0; k < frames; k++) exchange.Submit(k);
        exchange.Finish();
        dxrs::ThrowIfFailed(pt_synchronize(ctx), ctx, "pt_synchronize");
        if (rank == 0) {
            std::vector<dxrs::Float4> frame(size_t(w) * h);
            dxrs::ThrowIfFailed(pt_download(ctx, exchange.Frame((frames - 1) % batch), frame.data(), frame.size() * sizeof(dxrs::Float4)), ctx, "pt_download");
            PtStats totals{};
            dxrs::ThrowIfFailed(pt_get_totals(ctx, &totals, 0), ctx, "pt_get_totals");
            std::printf("rank 0 of %u: root tiles %u, tiles per other rank %u, rays (this rank) %llu\n", world, exchange.RootTiles(), exchange.TilesPerOtherRank(),
                        (unsigned long long)totals.rays);
            FILE* f = std::fopen(argv[8], "wb");
            if (!f) return 3;
            std::fwrite(frame.data(), sizeof(dxrs::Float4), frame.size(), f);
            std::fclose(f);
        }
        dxrs::ThrowIfFailed(pt_comm_destroy(ctx), ctx, "pt_comm_destroy");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
