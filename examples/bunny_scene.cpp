// examples/bunny_scene.cpp -- the reference's BunnyScene (myapp.cpp:13-52) through the C++ host adapter: an OBJ model
// loaded with TriangleMesh::LoadObj semantics under Translate(.25,-1.05,.5) * RotateY(180 deg), the curved backdrop,
// key / fill / back sphere lights and the sky; a few progressive frames, an orbit step of the RotatingCamera
// (myapp.cpp:143-149: accumulator cleared on camera motion) and PNG / PFM output.
//
//   g++ -std=c++17 -Iinclude examples/bunny_scene.cpp -o bunny_scene ag-pathtracer_amd/libagpt_hip.so
//   ./bunny_scene model.obj out_prefix [spp]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "agpt_host.hpp"

using namespace agpt;

static float lin(float c) { return std::pow(c, 2.2f); }                       // rgb2lin, template/common.h:29-32
static float3 rgb2lin(float3 c) { return float3{lin(c.x), lin(c.y), lin(c.z)}; }
static float3 hex2lin(int h) { return rgb2lin(float3{((h >> 16) & 0xFF) / 255.f, ((h >> 8) & 0xFF) / 255.f, (h & 0xFF) / 255.f}); }

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s model.obj out_prefix [spp]\n", argv[0]);
        return 2;
    }
    const std::string obj = argv[1], prefix = argv[2];
    const int spp = argc > 3 ? std::atoi(argv[3]) : 16, W = 320, H = 320;
    try {
        Context ctx(0);
        Scene scene(ctx);
        const int cute = DisneyMaterial::Make(scene, hex2lin(0xc5b5d2), .25f, 0.f);
        const int floor = DisneyMaterial::Make(scene, hex2lin(0xcbceb1), 1.f, 0.f);
        scene.primitives_push_back(TriangleMesh::CreateBackdrop(float3{0, -1, 20}, float3{40, 20, 40}, 7.5f, 32), floor, 1);
        // mat4::Translate(.25, -1.05, .5) * mat4::RotateY(radians(180)), row-major (template/precomp.h:875-876)
        const float a = 3.14159265358979323846f, c = std::cos(a), s = std::sin(a);
        const float transform[16] = {c, 0, s, .25f, 0, 1, 0, -1.05f, -s, 0, c, .5f, 0, 0, 0, 1};
        const TriangleMesh model = TriangleMesh::LoadObj(obj, transform);
        scene.primitives_push_back(model, cute, 1);
        const float n = std::sqrt(1.24f * 1.24f + 0.1f * 0.1f + 4.84f * 4.84f);
        scene.camera = CameraDesc{{1.24f / n * 5, 0.1f / n * 5, -4.84f / n * 5}, {.5f, 0, .5f}, {0, 1, 0}, 1.f, 30.f, 0.f};
        const float3 white{1.f, .941f, .914f};
        scene.addAreaLight(Sphere{float3{0, 25, -20}, 1.f}, float3{white.x * 200, white.y * 200, white.z * 200});   // key
        scene.addAreaLight(Sphere{float3{10, 25, -20}, 1.f}, float3{white.x * 50, white.y * 50, white.z * 50});     // fill
        scene.addAreaLight(Sphere{float3{0, 20, 10}, 5.f}, white);                                                  // back
        scene.lights_push_back(UniformInfiniteLight{float3{.4f, .45f, .5f}});
        scene.commit();

        PathTracer integrator;
        Accumulator accumulator(ctx, W, H);
        RotatingCamera orbit(scene.camera);
        unsigned long long rays = 0;
        for (int view = 0; view < 2; view++) {
            if (view == 1) {  // MyApp::Tick after a mouse drag: new lookfrom, accumulator cleared
                scene.camera = orbit.update(-0.15f, 0.6f);
                scene.set_camera();
                accumulator.Clear();
            }
            for (int f = 0; f < spp; f++) {   // progressive: one sample per pixel per Tick
                const agpt_stats st = integrator.Render(scene, accumulator, 1);
                rays += st.closest_rays + st.anyhit_rays;
            }
            const std::string png = prefix + "_view" + std::to_string(view) + ".png";
            const std::vector<uint32_t> rgb = accumulator.CopyToSurface();
            check(agpt_write_png(png.c_str(), rgb.data(), W, H), "agpt_write_png");
            if (view == 0) {
                const std::vector<float> px = accumulator.Download();
                check(agpt_write_pfm((prefix + "_view0.pfm").c_str(), px.data(), W, H, accumulator.NumSamples()), "agpt_write_pfm");
                double mean[3] = {0, 0, 0};
                for (size_t i = 0; i < (size_t)W * H; i++)
                    for (int k = 0; k < 3; k++) mean[k] += px[4 * i + k];
                std::printf("view0 spp=%d triangles=%zu mean_rgb=%.6f %.6f %.6f\n", accumulator.NumSamples(), model.indices.size() / 9,
                            mean[0] / (W * H) / spp, mean[1] / (W * H) / spp, mean[2] / (W * H) / spp);
            }
        }
        std::printf("rays=%llu lookfrom_after_orbit=%.6f %.6f %.6f\n", rays, scene.camera.lookfrom[0], scene.camera.lookfrom[1],
                    scene.camera.lookfrom[2]);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
