// host_shim.cpp — C entry points over include/rustray_host.hpp so that the C++ host layer (Camera, RaytracingConfig,
// Raytracing, RendererManager) can be driven from the Python tests through ctypes.  Test plumbing, not part of the ABI.
#include "../../include/rustray_host.hpp"

#include <chrono>
#include <thread>

using namespace rustray;

static Camera make_camera(float fov, const float* eye, const float* up, const float* dir, float cnear, float cfar) {
    Camera c;
    c.fov = fov;
    c.eye_pos = Vec3{eye[0], eye[1], eye[2]}; c.up = Vec3{up[0], up[1], up[2]}; c.dir = Vec3{dir[0], dir[1], dir[2]};
    c.clipping_near = cnear; c.clipping_far = cfar;
    return c;
}

extern "C" int rh_camera(float fov, const float* eye, const float* up, const float* dir, float cnear, float cfar, uint32_t w, uint32_t h,
                         float* proj_inv16, float* view_inv16, int* is_default, const float* point, int* point_in_frustum) {
    Camera c = make_camera(fov, eye, up, dir, cnear, cfar);
    c.init(w, h);
    const rr_camera rc = c.c_struct();
    for (int i = 0; i < 16; i++) { proj_inv16[i] = rc.projection_inverse[i]; view_inv16[i] = rc.view_inverse[i]; }
    *is_default = c.is_default_cam() ? 1 : 0;
    if (point && point_in_frustum) *point_in_frustum = c.is_point_in_frustum(Vec3{point[0], point[1], point[2]}) ? 1 : 0;
    return 0;
}

static RaytracingConfig from_c(const rr_config* c) {
    RaytracingConfig r;
    r.monte_carlo = c->monte_carlo != 0; r.samples = c->samples; r.focal_length = c->focal_length; r.aperture_size = c->aperture_size;
    r.fog_density = c->fog_density; r.fog_color = Vec3{c->fog_color[0], c->fog_color[1], c->fog_color[2]};
    r.max_recursion = c->max_recursion; r.gamma_correction = c->gamma_correction != 0; r.seed = c->seed;
    return r;
}

// base.apply(n) -> out
extern "C" void rh_config_apply(const rr_config* base, const rr_config* n, rr_config* out) {
    RaytracingConfig b = from_c(base);
    b.apply(from_c(n));
    *out = b.c_struct();
}

// One frame through RendererManager::start / is_done / stop.  stats: [passes, rendered pixels, is_done, drained pixels,
// elapsed ms, was running right after start].  stop_after_passes > 0 calls stop() once that many passes were seen.
extern "C" int rh_render(const rr_flat_scene* fs, int device, float fov, const float* eye, const float* up, const float* dir, float cnear, float cfar,
                         const rr_config* cfg, uint32_t w, uint32_t h, uint32_t min_passes, int stop_after_passes,
                         uint8_t* rgba, float* normal, float* depth, uint32_t* ids, uint64_t* stats, int pick_x, int pick_y, float* pick_out) {
    auto scene = std::make_shared<DeviceScene>(*fs, device);
    if (!scene->ok()) return -1;
    auto rt = std::make_shared<Raytracing>(scene);
    rt->camera = make_camera(fov, eye, up, dir, cnear, cfar);
    rt->config.apply(from_c(cfg));
    rt->config.seed = cfg->seed;
    RendererManager mgr((int32_t)w, (int32_t)h, rt);
    mgr.min_passes = min_passes;
    mgr.start();
    stats[5] = mgr.is_running() ? 1 : 0;
    uint64_t drained = 0;
    while (mgr.is_running()) {
        drained += mgr.drain([](const PixelData&) {});
        if (stop_after_passes > 0 && mgr.passes() >= (uint32_t)stop_after_passes) mgr.stop();
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    mgr.wait();
    std::vector<uint8_t> im; std::vector<float> nr, dp; std::vector<uint32_t> id;
    uint64_t last = 0;
    last = mgr.drain([&](const PixelData& p) { (void)p; });
    drained += last;
    mgr.frame(&im, &nr, &dp, &id);
    std::memcpy(rgba, im.data(), im.size());
    std::memcpy(normal, nr.data(), nr.size() * 4);
    std::memcpy(depth, dp.data(), dp.size() * 4);
    std::memcpy(ids, id.data(), id.size() * 4);
    stats[0] = mgr.passes(); stats[1] = mgr.get_rendered_pixels(); stats[2] = mgr.is_done() ? 1 : 0; stats[3] = drained;
    stats[4] = mgr.check_and_get_elapsed_time();
    if (stop_after_passes == -1) { // the caller wants the post-processed image (cavity + outline)
        const std::vector<uint8_t> pp = mgr.post_processing(true, true, device);
        std::memcpy(rgba, pp.data(), pp.size());
    }
    if (pick_out) {
        auto p = rt->pick(pick_x, pick_y);
        pick_out[0] = p ? 1.0f : 0.0f; pick_out[1] = p ? (float)p->first : 0.0f; pick_out[2] = p ? p->second : 0.0f;
    }
    return mgr.failed() ? -2 : 0;
}

// Animation::frame_transforms for a two-keyframe animation of ONE named object among `n_items` items
// (names "item0", "item1", ...; the animated one is `animated`): tests the keyframe selection, the interpolation and
// the transformation order against the Python mirror.
extern "C" int rh_animation_frame(uint32_t fps, uint64_t t1_ms, const float* tr0, const float* rot0, const float* sc0,
                                  const float* tr1, const float* rot1, const float* sc1, uint32_t n_items, uint32_t animated,
                                  uint64_t frame, float* trans, float* trans_inv, uint64_t* frames_amount, int* exists) {
    Animation an;
    an.enabled = true; an.fps = fps;
    const std::string name = "item" + std::to_string(animated);
    Keyframe k0; k0.time = 0;
    k0.objects.push_back(Frame{name, Vec3{tr0[0], tr0[1], tr0[2]}, Vec3{rot0[0], rot0[1], rot0[2]}, Vec3{sc0[0], sc0[1], sc0[2]}});
    Keyframe k1; k1.time = t1_ms;
    k1.objects.push_back(Frame{name, Vec3{tr1[0], tr1[1], tr1[2]}, Vec3{rot1[0], rot1[1], rot1[2]}, Vec3{sc1[0], sc1[1], sc1[2]}});
    an.keyframes = {k0, k1};
    std::vector<std::string> names;
    for (uint32_t i = 0; i < n_items; i++) names.push_back("item" + std::to_string(i));
    *frames_amount = an.get_frames_amount_to_render();
    *exists = an.frame_exists(frame) ? 1 : 0;
    return an.frame_transforms(names, frame, trans, trans_inv) ? 1 : 0;
}
