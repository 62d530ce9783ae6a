// pt_host_c.cpp -- C exports of the host-side mirror (scene generators, camera controller, Halton jitter)
// so that Python (tests/, bench.py) drives exactly the C++ host logic a C++ application would use.
// Pure host code: no HIP, no oracle.
#include <cstring>

#include "Camera.hpp"
#include "HaltonSampler.hpp"
#include "MyScene.hpp"
#include "Random.hpp"

using namespace dxrs;

namespace {

int export_scene(const SceneDesc& desc, PtSphere* spheres, PtMaterial* materials, uint32_t capacity, uint32_t* count, PtSceneData* sd)
{
    Scene scene;
    scene.Load(desc);
    const uint32_t n = scene.GetObjectCount();
    if (count) *count = n;
    if (sd) *sd = scene.GetSceneData();
    if (!spheres || !materials) return 0;  // size query
    if (capacity < n) return 1;
    std::memcpy(spheres, scene.GetSpheres().data(), n * sizeof(PtSphere));
    std::memcpy(materials, scene.GetMaterials().data(), n * sizeof(PtMaterial));
    return 0;
}

}  // namespace

extern "C" {

// kind 0: demo default scene (MySceneDesc); 1: 16-sphere scene (SmallSceneDesc); 2: procedural (count spheres + ground)
int pth_scene(uint32_t kind, uint32_t seed, uint32_t count_param, PtSphere* spheres, PtMaterial* materials, uint32_t capacity,
              uint32_t* count, PtSceneData* scene_data)
{
    switch (kind) {
        case 0: return export_scene(MySceneDesc(seed), spheres, materials, capacity, count, scene_data);
        case 1: return export_scene(SmallSceneDesc(seed), spheres, materials, capacity, count, scene_data);
        case 2: return export_scene(ProceduralSceneDesc(count_param, seed), spheres, materials, capacity, count, scene_data);
        default: return 2;
    }
}

// CameraController at `position` with identity rotation (or looking at `look_at` when non-null), SetLens(hfov, w/h),
// jitter = Halton2D(jitter_index + 1) - 0.5 when jitter_enabled (Source/App.cpp:542-551), jitter_index cycling mod jitter_count.
void pth_camera(const float position[3], const float* look_at, float hfov, uint32_t width, uint32_t height, int jitter_enabled,
                uint32_t jitter_index, uint32_t jitter_count, PtCamera* out)
{
    CameraController controller;
    controller.SetPosition({ position[0], position[1], position[2] });
    if (look_at) controller.LookAt({ look_at[0], look_at[1], look_at[2] }, { 0, 1, 0 }, false);
    controller.SetLens(hfov, static_cast<float>(width) / static_cast<float>(height));
    Float2 jitter{};
    if (jitter_enabled) {
        const auto h = HaltonSampler::Get2D(jitter_index % (jitter_count ? jitter_count : 1) + 1);
        jitter = { h.x - 0.5f, h.y - 0.5f };
    }
    Camera camera;
    controller.Fill(camera, jitter);
    camera.PreviousPosition = camera.Position;
    *out = ToPt(camera);
}

// The demo scene at simulation time `time` seconds (closed-form motion, MyScene::SetTime): spheres only (the materials and
// the object order are those of pth_scene(0, seed)).
int pth_scene_at_time(uint32_t seed, double time, PtSphere* spheres, uint32_t capacity, uint32_t* count)
{
    MyScene scene(seed);
    scene.SetTime(time);
    const uint32_t n = scene.GetObjectCount();
    if (count) *count = n;
    if (!spheres) return 0;
    if (capacity < n) return 1;
    std::memcpy(spheres, scene.GetSpheres().data(), n * sizeof(PtSphere));
    return 0;
}

float pth_halton(uint32_t index, uint32_t base) { return Halton(index, base); }

// raw draws of the scene-generation RNG (Random.ixx mirror), for the host-logic tests
void pth_random_floats(uint32_t seed, uint32_t n, float* out)
{
    Random random(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = random.Float();
}

}  // extern "C"
