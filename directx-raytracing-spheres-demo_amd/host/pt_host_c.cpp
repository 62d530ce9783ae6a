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

// The demo scene with its textured objects (MySceneDesc(seed, true)) at simulation time `time`: texture table, per-object
// maps and world-space rotations.  Two-step protocol: call with null arrays for the counts, then with storage.
//   image_info[4 * t] = { width, height, format (PtTexture::Format), byte offset into `pixels` }
// flags: bit 0 = the textured objects, bit 1 = the lat-long environment map (MySceneDesc's `environmentMap`); scene_data (may
// be null) receives the scene's SceneData (EnvironmentLightTextureDescriptor / Transform for bit 1).
int pth_demo_textures_ex(uint32_t seed, double time, uint32_t flags, uint32_t* n_textures, uint32_t* n_objects, uint64_t* pixel_bytes,
                         uint32_t* image_info, uint8_t* pixels, PtObjectTextures* object_textures, float* rotations, PtSceneData* scene_data)
{
    MyScene scene(seed, (flags & 1u) != 0, (flags & 2u) != 0);
    scene.SetTime(time);
    if (scene_data) *scene_data = scene.GetSceneData();
    const auto& tex = scene.GetTextures();
    uint64_t bytes = 0;
    for (const auto& t : tex) bytes += (t.ByteSize() + 15u) & ~size_t(15);
    if (n_textures) *n_textures = static_cast<uint32_t>(tex.size());
    if (n_objects) *n_objects = scene.GetObjectCount();
    if (pixel_bytes) *pixel_bytes = bytes;
    if (!image_info || !pixels || !object_textures || !rotations) return 0;
    uint64_t off = 0;
    for (size_t t = 0; t < tex.size(); t++) {
        image_info[4 * t] = tex[t].Width; image_info[4 * t + 1] = tex[t].Height;
        image_info[4 * t + 2] = tex[t].Format();
        image_info[4 * t + 3] = static_cast<uint32_t>(off);
        std::memcpy(pixels + off, tex[t].Data(), tex[t].ByteSize());
        off += (tex[t].ByteSize() + 15u) & ~size_t(15);
    }
    std::memcpy(object_textures, scene.GetObjectTextures().data(), scene.GetObjectCount() * sizeof(PtObjectTextures));
    std::memcpy(rotations, scene.GetRotations().data(), scene.GetRotations().size() * sizeof(float));
    return 0;
}

int pth_demo_textures(uint32_t seed, double time, uint32_t* n_textures, uint32_t* n_objects, uint64_t* pixel_bytes,
                      uint32_t* image_info, uint8_t* pixels, PtObjectTextures* object_textures, float* rotations)
{
    return pth_demo_textures_ex(seed, time, 1u, n_textures, n_objects, pixel_bytes, image_info, pixels, object_textures, rotations, nullptr);
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
