// spheres.cpp — the C++ host layer end to end: build a small scene, render it with RendererManager the way the
// reference's main loop drives its renderer (start, poll is_done, drain pixels; reference src/run.rs, src/renderer.rs),
// write a PPM.  Needs a MI355X at run time; `make -C rustray_amd/csrc` only checks that it builds.
//   examples/spheres [out.ppm] [width height samples]
#include "../include/rustray_host.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

using namespace rustray;

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "spheres.ppm";
    const int w = argc > 3 ? atoi(argv[2]) : 640, h = argc > 3 ? atoi(argv[3]) : 360;
    const int spp = argc > 4 ? atoi(argv[4]) : 64;
    if (rr_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", rr_last_error()); return 2; }

    FlatScene flat;
    Material floor; floor.base_color = {0.7f, 0.7f, 0.75f}; floor.reflectivity = 0.25f;
    flat.add_mesh({-20, -1, 20, 20, -1, 20, 20, -1, -20, -20, -1, -20}, {0, 1, 2, 0, 2, 3}, floor, 1);
    Material red; red.base_color = {0.9f, 0.15f, 0.1f};
    Material glass; glass.base_color = {0.9f, 0.95f, 1.0f}; glass.alpha = 0.2f; glass.refraction_index = 1.5f; glass.reflectivity = 0.3f;
    Material mirror; mirror.base_color = {0.9f, 0.9f, 0.9f}; mirror.reflectivity = 0.8f; mirror.roughness = 0.02f;
    flat.add_sphere({-2.2f, 0.0f, -6.0f}, 1.0f, red, 2);
    flat.add_sphere({0.0f, 0.0f, -5.0f}, 1.0f, glass, 3);
    flat.add_sphere({2.2f, 0.0f, -6.0f}, 1.0f, mirror, 4);
    flat.add_point_light({-2.0f, 10.0f, 5.0f}, 200.0f); // the reference's default light (src/scene.rs:1386-1401)

    const rr_flat_scene fs = flat.c_struct();
    auto scene = std::make_shared<DeviceScene>(fs, 0);
    if (!scene->ok()) { fprintf(stderr, "scene: %s\n", scene->error().c_str()); return 1; }
    auto rt = std::make_shared<Raytracing>(scene);
    rt->camera.eye_pos = {0.0f, 0.6f, 1.5f};
    rt->camera.dir = {0.0f, -0.12f, -1.0f};
    RaytracingConfig cfg; cfg.samples = (uint16_t)spp; cfg.monte_carlo = true;
    rt->config.apply(cfg);

    RendererManager mgr(w, h, rt);
    mgr.start();
    uint32_t seen = 0;
    while (mgr.is_running()) {
        if (mgr.passes() != seen) { seen = mgr.passes(); fprintf(stderr, "pass %u: %llu of %d pixels\n", seen, (unsigned long long)mgr.get_rendered_pixels(), w * h); }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    mgr.wait();
    if (mgr.failed()) { fprintf(stderr, "render: %s\n", mgr.last_error().c_str()); return 1; }
    fprintf(stderr, "done in %llu ms\n", (unsigned long long)mgr.check_and_get_elapsed_time());

    std::vector<uint8_t> rgba;
    mgr.frame(&rgba);
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) fwrite(&rgba[4 * i], 1, 3, f);
    fclose(f);
    auto hit = rt->pick(w / 2, h / 2);
    if (hit) fprintf(stderr, "pick at the centre: object %u at distance %.3f\n", hit->first, hit->second);
    return 0;
}
