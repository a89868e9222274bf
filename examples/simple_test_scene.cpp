// examples/simple_test_scene.cpp -- the reference's SimpleTestScene geometry (myapp.cpp:55-114) rendered through the
// C++ host adapter: what MyApp::Init + a few MyApp::Tick calls look like on top of libagpt_hip.so.
//
//   g++ -std=c++17 -Iinclude examples/simple_test_scene.cpp -o simple_test_scene ag-pathtracer_amd/libagpt_hip.so
//   ./simple_test_scene [frames] [environment.hdr]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "agpt_host.hpp"

using namespace agpt;

static float3 hex2lin(int h) {  // template/common.h:29-39
    return float3{std::pow(((h >> 16) & 0xFF) / 255.f, 2.2f), std::pow(((h >> 8) & 0xFF) / 255.f, 2.2f),
                  std::pow((h & 0xFF) / 255.f, 2.2f)};
}

int main(int argc, char** argv) {
    const int W = 256, H = 256, frames = argc > 1 ? std::atoi(argv[1]) : 4;
    try {
        Context ctx(0);
        Scene scene(ctx);
        // --- MyApp::Init / SimpleTestScene ---------------------------------------------------------------
        int gold = DisneyMaterial::Make(scene, float3{0.944f, 0.776f, 0.373f}, .5f, 1.f);
        int floor = DisneyMaterial::Make(scene, hex2lin(0xcbceb1), 1.f, 0.f);
        TriangleMesh backdrop = TriangleMesh::CreateBackdrop(float3{0, -1, 20}, float3{40, 20, 40}, 7.5f, 32);
        scene.primitives_push_back(backdrop, floor, 1);                    // make_shared<BVHTriMesh>(backdrop, floor, 1)
        scene.primitives_push_back(Sphere{float3{0, 0, 0}, 1.f}, gold);    // make_shared<Sphere>(..., gold)
        scene.addAreaLight(Sphere{float3{0, 25, -20}, 1.f}, float3{200.f, .941f * 200, .914f * 200});
        if (argc > 2)   // myapp.cpp:113: scene->lights.push_back(make_shared<InfiniteAreaLight>("small_workshop_1k.hdr"))
            scene.lights_push_back(InfiniteAreaLight(argv[2]));
        else
            scene.lights_push_back(UniformInfiniteLight{float3{.4f, .45f, .5f}});
        scene.camera = CameraDesc{{-1.46f, 1.16f, -4.64f}, {0, 0, 0}, {0, 1, 0}, 1.f, 45.f, 0.f};
        scene.commit();
        PathTracer integrator;  // MaxDepth = 5
        Accumulator accumulator(ctx, W, H);
        // --- MyApp::Tick, `frames` times: one sample per pixel per frame ------------------------------------
        unsigned long long rays = 0;
        for (int f = 0; f < frames; f++) {
            agpt_stats st = integrator.Render(scene, accumulator, 1);
            rays += st.closest_rays + st.anyhit_rays;
        }
        std::vector<float> px = accumulator.Download();
        double mean[3] = {0, 0, 0};
        for (size_t i = 0; i < (size_t)W * H; i++)
            for (int c = 0; c < 3; c++) mean[c] += px[4 * i + c];
        std::vector<uint32_t> rgb = accumulator.CopyToSurface();
        std::printf("frames=%d rays=%llu mean_rgb=%.6f %.6f %.6f first_pixel=0x%06x\n", accumulator.NumSamples(), rays,
                    mean[0] / (W * H) / frames, mean[1] / (W * H) / frames, mean[2] / (W * H) / frames, rgb[0]);
        agpt_ray pick{{-1.46f, 1.16f, -4.64f}, {1.46f, -1.16f, 4.64f}, 3.402823466e+38f};
        agpt_hit hit;
        if (scene.Intersect(pick, hit)) std::printf("pick: prim=%d t=%.6f\n", hit.prim, hit.t);
        // one Integrator::Li call on that ray, drawing from the reference's global stream (seed: template/template.cpp:667)
        uint32_t random_state = 0x12345678u;
        const float3 L = integrator.Li(scene, pick, random_state);
        std::printf("li: %.9g %.9g %.9g state=0x%08x\n", L.x, L.y, L.z, random_state);
        // the reference's other integrator, the uv view (integrator.h:107-118)
        const float3 uvw = DbgIntegrator().Li(scene, pick);
        std::printf("dbg: %.9g %.9g %.9g\n", uvw.x, uvw.y, uvw.z);
    } catch (const Error& e) {
        std::fprintf(stderr, "agpt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
