// rustray_host.hpp — C++17 host layer over the C ABI of rustray_hip.h.
//
// The reference's host code around the trace loop is compiled Rust; no Rust toolchain exists in this pipeline, so
// this header restates, in C++, the part of that host code which drives the path: `Camera` (reference
// src/camera.rs:18-140), `RaytracingConfig` / `PixelData` / `Raytracing` (src/raytracing.rs:57-273) and
// `RendererManager` (src/renderer.rs:38-251) with the same names, argument meaning and call surface.  What changes
// behind that surface: `RendererManager::start` does not spawn num_cpus-2 workers over a queue of 2x2-pixel cells
// (src/renderer.rs:105-172) but ONE thread that issues a frame-level call into librustray_hip.so
// (rr_render_progressive); finished pixels arrive per pass (every pixel, more samples) instead of cell by cell.
//
// Header-only; link against librustray_hip.so.  Nothing here touches the GPU directly.
#pragma once

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rustray_hip.h"

namespace rustray {

// helper::approx_equal (reference src/helper.rs:11-20): six truncated decimals, in f32
inline bool approx_equal(float a, float b) {
    const float f = 1000000.0f;
    return std::trunc(a * f) == std::trunc(b * f);
}

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};

// reference src/raytracing.rs:57-72
struct PixelData {
    uint8_t r = 0, g = 0, b = 0;
    Vec3 normal;
    float depth = 0.0f;
    uint32_t object_id = 0;
    int32_t x = 0, y = 0;
};

// reference src/raytracing.rs:92-185
struct RaytracingConfig {
    bool monte_carlo = false;
    uint16_t samples = 1; // this includes anti aliasing
    float focal_length = 1.0f;
    float aperture_size = 1.0f; // 1 means off
    float fog_density = 0.0f;
    Vec3 fog_color{0.4f, 0.4f, 0.4f};
    uint16_t max_recursion = 6;
    bool gamma_correction = false;
    uint64_t seed = 0; // not in the reference: its jitter draws from an un-seeded thread_rng

    // RaytracingConfig::apply: only fields that differ from the defaults are taken over (src/raytracing.rs:129-185)
    void apply(const RaytracingConfig& n) {
        const RaytracingConfig d;
        if (d.monte_carlo != n.monte_carlo) monte_carlo = n.monte_carlo;
        if (d.samples != n.samples) samples = n.samples;
        if (!approx_equal(d.focal_length, n.focal_length)) focal_length = n.focal_length;
        if (!approx_equal(d.aperture_size, n.aperture_size)) aperture_size = n.aperture_size;
        if (!approx_equal(d.fog_density, n.fog_density)) fog_density = n.fog_density;
        if (!approx_equal(d.fog_color.x, n.fog_color.x) || !approx_equal(d.fog_color.y, n.fog_color.y) || !approx_equal(d.fog_color.z, n.fog_color.z))
            fog_color = n.fog_color;
        if (d.max_recursion != n.max_recursion) max_recursion = n.max_recursion;
        if (d.gamma_correction != n.gamma_correction) gamma_correction = n.gamma_correction;
    }

    rr_config c_struct() const {
        rr_config c;
        std::memset(&c, 0, sizeof c);
        c.seed = seed;
        c.focal_length = focal_length; c.aperture_size = aperture_size; c.fog_density = fog_density;
        c.fog_color[0] = fog_color.x; c.fog_color[1] = fog_color.y; c.fog_color[2] = fog_color.z;
        c.samples = samples; c.max_recursion = max_recursion;
        c.monte_carlo = monte_carlo ? 1 : 0; c.gamma_correction = gamma_correction ? 1 : 0;
        return c;
    }
};

// reference src/camera.rs:10-15
constexpr float DEFAULT_FOV = 90.0f;
constexpr float DEFAULT_CLIPPING_NEAR = 0.001f;
constexpr float DEFAULT_CLIPPING_FAR = 1000.0f;

// reference src/camera.rs:18-140.  The trace loop consumes width, height and the two inverse matrices only
// (src/raytracing.rs:282-283, :349, :355, :369-393).  Matrices are column-major like nalgebra's; they are evaluated in
// double and rounded to f32 once, exactly as rustray_amd/camera.py does, so both host mirrors hand the library the
// same bits (the camera is harness code, outside the parity contract of the trace loop).
class Camera {
public:
    uint32_t width = 0, height = 0;
    float aspect_ratio = 0.0f;
    float fov = (float)(90.0 * 3.14159265358979323846 / 180.0);
    Vec3 eye_pos{0.0f, 0.0f, 0.0f}, up{0.0f, 1.0f, 0.0f}, dir{0.0f, 0.0f, -1.0f};
    float clipping_near = DEFAULT_CLIPPING_NEAR, clipping_far = DEFAULT_CLIPPING_FAR;
    double projection[16], view[16], projection_inverse[16], view_inverse[16]; // column-major

    Camera() {
        fov = (float)((double)DEFAULT_FOV * 3.14159265358979323846 / 180.0);
        identity(projection); identity(view); identity(projection_inverse); identity(view_inverse);
    }

    void init(uint32_t w, uint32_t h) { // src/camera.rs:69-77
        width = w; height = h;
        aspect_ratio = (float)w / (float)h;
        init_matrices();
    }

    void init_matrices() { // src/camera.rs:79-90: Perspective3::new, Isometry3::look_at_rh and their inverses
        const double a = aspect_ratio, fovy = fov, zn = clipping_near, zf = clipping_far;
        const double t = std::tan(fovy / 2.0);
        zero(projection);
        at(projection, 0, 0) = 1.0 / (a * t);
        at(projection, 1, 1) = 1.0 / t;
        at(projection, 2, 2) = (zf + zn) / (zn - zf);
        at(projection, 2, 3) = 2.0 * zf * zn / (zn - zf);
        at(projection, 3, 2) = -1.0;
        zero(projection_inverse);
        at(projection_inverse, 0, 0) = 1.0 / at(projection, 0, 0);
        at(projection_inverse, 1, 1) = 1.0 / at(projection, 1, 1);
        at(projection_inverse, 2, 3) = -1.0;
        at(projection_inverse, 3, 2) = 1.0 / at(projection, 2, 3);
        at(projection_inverse, 3, 3) = at(projection, 2, 2) / at(projection, 2, 3);
        const double e[3] = {eye_pos.x, eye_pos.y, eye_pos.z};
        const double tg[3] = {e[0] + (double)dir.x, e[1] + (double)dir.y, e[2] + (double)dir.z};
        double z[3] = {e[0] - tg[0], e[1] - tg[1], e[2] - tg[2]};
        normalize(z);
        const double u[3] = {up.x, up.y, up.z};
        double x[3]; cross(u, z, x); normalize(x);
        double y[3]; cross(z, x, y);
        identity(view);
        for (int k = 0; k < 3; k++) { at(view, 0, k) = x[k]; at(view, 1, k) = y[k]; at(view, 2, k) = z[k]; }
        at(view, 0, 3) = -dot(x, e); at(view, 1, 3) = -dot(y, e); at(view, 2, 3) = -dot(z, e);
        identity(view_inverse);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) at(view_inverse, r, c) = at(view, c, r);
        for (int r = 0; r < 3; r++) at(view_inverse, r, 3) = e[r];
    }

    bool is_default_cam() const { // src/camera.rs:92-123
        return approx_equal(eye_pos.x, 0.0f) && approx_equal(eye_pos.y, 0.0f) && approx_equal(eye_pos.z, 0.0f) &&
               approx_equal(dir.x, 0.0f) && approx_equal(dir.y, 0.0f) && approx_equal(dir.z, -1.0f) &&
               approx_equal(up.x, 0.0f) && approx_equal(up.y, 1.0f) && approx_equal(up.z, 0.0f) &&
               approx_equal(fov, (float)((double)DEFAULT_FOV * 3.14159265358979323846 / 180.0)) &&
               approx_equal(clipping_near, DEFAULT_CLIPPING_NEAR) && approx_equal(clipping_far, DEFAULT_CLIPPING_FAR);
    }

    void set_cam_position(Vec3 eye, Vec3 d) { eye_pos = eye; dir = d; init_matrices(); } // src/camera.rs:125-131

    bool is_point_in_frustum(Vec3 p) const { // src/camera.rs:133-140
        double pv[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += cat(projection, r, k) * cat(view, k, c);
            pv[c * 4 + r] = s;
        }
        const double h[4] = {p.x, p.y, p.z, 1.0};
        double o[4];
        for (int r = 0; r < 4; r++) { o[r] = 0.0; for (int k = 0; k < 4; k++) o[r] += pv[k * 4 + r] * h[k]; }
        return std::fabs(o[0]) <= o[3] && std::fabs(o[1]) <= o[3] && std::fabs(o[2]) <= o[3];
    }

    rr_camera c_struct() const {
        rr_camera c;
        c.width = width; c.height = height;
        for (int i = 0; i < 16; i++) { c.projection_inverse[i] = (float)projection_inverse[i]; c.view_inverse[i] = (float)view_inverse[i]; }
        return c;
    }

private:
    static double& at(double* m, int r, int c) { return m[c * 4 + r]; }
    static double cat(const double* m, int r, int c) { return m[c * 4 + r]; }
    static void zero(double* m) { for (int i = 0; i < 16; i++) m[i] = 0.0; }
    static void identity(double* m) { zero(m); m[0] = m[5] = m[10] = m[15] = 1.0; }
    static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static void cross(const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    }
    static void normalize(double* v) { const double n = std::sqrt(dot(v, v)); v[0] /= n; v[1] /= n; v[2] /= n; }
};


// A flat scene under construction: owns the arrays an rr_flat_scene points to.  Stands in for what the Rust shim of
// INTEGRATION.md produces from `Scene` (items in Scene.items order, one texture-less material-cache entry per item).
struct Material { // defaults of Material::new, reference src/shape/mod.rs:138-180
    Vec3 ambient_color{0.0f, 0.0f, 0.0f}, base_color{1.0f, 1.0f, 1.0f}, specular_color{0.8f, 0.8f, 0.8f};
    float alpha = 1.0f, shininess = 150.0f, reflectivity = 0.0f, refraction_index = 1.0f, normal_map_strength = 1.0f;
    float shadow_softness = 0.01f, roughness = 0.0f;
    int32_t texture[RR_TEX_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1};
    bool texture_filtering_nearest = false, cast_shadow = true, receive_shadow = true, monte_carlo = true;
    bool smooth_shading = true, reflection_only = false, backface_cullig = true;

    rr_material c_struct(bool with_textures) const {
        rr_material m;
        std::memset(&m, 0, sizeof m);
        const Vec3* src[3] = {&ambient_color, &base_color, &specular_color};
        float* dst[3] = {m.ambient_color, m.base_color, m.specular_color};
        for (int k = 0; k < 3; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
        m.alpha = alpha; m.shininess = shininess; m.reflectivity = reflectivity; m.refraction_index = refraction_index;
        m.normal_map_strength = normal_map_strength; m.shadow_softness = shadow_softness; m.roughness = roughness;
        for (int k = 0; k < RR_TEX_COUNT; k++) m.texture[k] = with_textures ? texture[k] : -1;
        m.texture_filtering_nearest = texture_filtering_nearest; m.cast_shadow = cast_shadow; m.receive_shadow = receive_shadow;
        m.monte_carlo = monte_carlo; m.smooth_shading = smooth_shading; m.reflection_only = reflection_only; m.backface_cullig = backface_cullig;
        return m;
    }
};

class FlatScene {
public:
    // a sphere of `radius` centred at `c` (Sphere::new + translation, reference src/shape/sphere.rs, src/shape/mod.rs:708-729)
    uint32_t add_sphere(Vec3 c, float radius, const Material& mat, uint32_t id) {
        rr_item it;
        std::memset(&it, 0, sizeof it);
        it.kind = RR_ITEM_SPHERE; it.id = id; it.mesh = -1; it.radius = radius;
        identity(it.trans); identity(it.trans_inv);
        it.trans[12] = c.x; it.trans[13] = c.y; it.trans[14] = c.z;
        it.trans_inv[12] = -c.x; it.trans_inv[13] = -c.y; it.trans_inv[14] = -c.z;
        for (int k = 0; k < 3; k++) { it.bbox_min[k] = -radius; it.bbox_max[k] = radius; }
        it.visible = 1;
        return push_item(it, mat);
    }
    // a triangle mesh in world space (identity transform); positions: 3 floats per vertex, indices: 3 per triangle
    uint32_t add_mesh(const std::vector<float>& positions, const std::vector<uint32_t>& indices, const Material& mat, uint32_t id) {
        mesh_pos_.push_back(positions); mesh_idx_.push_back(indices);
        rr_item it;
        std::memset(&it, 0, sizeof it);
        it.kind = RR_ITEM_MESH; it.id = id; it.mesh = (int32_t)mesh_pos_.size() - 1;
        identity(it.trans); identity(it.trans_inv);
        for (int k = 0; k < 3; k++) { it.bbox_min[k] = 3.0e38f; it.bbox_max[k] = -3.0e38f; }
        for (size_t v = 0; v + 2 < positions.size(); v += 3)
            for (int k = 0; k < 3; k++) { it.bbox_min[k] = std::fmin(it.bbox_min[k], positions[v + k]); it.bbox_max[k] = std::fmax(it.bbox_max[k], positions[v + k]); }
        it.visible = 1;
        return push_item(it, mat);
    }
    void add_point_light(Vec3 pos, float intensity, Vec3 color = Vec3{1.0f, 1.0f, 1.0f}) {
        rr_light l;
        std::memset(&l, 0, sizeof l);
        l.pos[0] = pos.x; l.pos[1] = pos.y; l.pos[2] = pos.z;
        l.color[0] = color.x; l.color[1] = color.y; l.color[2] = color.z;
        l.intensity = intensity; l.light_type = RR_LIGHT_POINT; l.enabled = 1;
        lights_.push_back(l);
    }
    // valid while this object lives and is not modified
    rr_flat_scene c_struct() {
        meshes_.clear();
        for (size_t i = 0; i < mesh_pos_.size(); i++) {
            rr_mesh m;
            std::memset(&m, 0, sizeof m);
            m.positions = mesh_pos_[i].data(); m.indices = mesh_idx_[i].data();
            m.n_vertices = (uint32_t)(mesh_pos_[i].size() / 3); m.n_triangles = (uint32_t)(mesh_idx_[i].size() / 3);
            meshes_.push_back(m);
        }
        rr_flat_scene fs;
        std::memset(&fs, 0, sizeof fs);
        fs.abi_version = RR_ABI_VERSION;
        fs.n_items = (uint32_t)items_.size(); fs.items = items_.data();
        fs.n_meshes = (uint32_t)meshes_.size(); fs.meshes = meshes_.data();
        fs.n_materials = (uint32_t)materials_.size(); fs.materials = materials_.data();
        fs.n_lights = (uint32_t)lights_.size(); fs.lights = lights_.data();
        return fs;
    }

private:
    static void identity(float* m) { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    uint32_t push_item(rr_item& it, const Material& mat) {
        materials_.push_back(mat.c_struct(true));  // get_material()
        materials_.push_back(mat.c_struct(false)); // get_material_cache_without_textures(), src/shape/mod.rs:769-772
        it.material = (int32_t)materials_.size() - 2; it.material_cache = (int32_t)materials_.size() - 1;
        items_.push_back(it);
        return (uint32_t)items_.size() - 1;
    }
    std::vector<rr_item> items_;
    std::vector<rr_material> materials_;
    std::vector<rr_light> lights_;
    std::vector<rr_mesh> meshes_;
    std::vector<std::vector<float>> mesh_pos_;
    std::vector<std::vector<uint32_t>> mesh_idx_;
};

// ---------------------------------------------------------------------------------------------------------
// Keyframe animation: reference src/animation.rs:9-205 and Scene::apply_frame (src/scene.rs:1695-1713).  Only item
// transforms change between frames, so a frame step on the device is rr_scene_update_transforms.
// ---------------------------------------------------------------------------------------------------------
struct Frame { // src/animation.rs:9-30; unset components are None in the reference
    std::string object_name;
    std::optional<Vec3> translation, rotation /* radians */, scale;
};
struct Keyframe { // src/animation.rs:35-52
    uint64_t time = 0; // milliseconds
    std::vector<Frame> objects;
};

class Animation {
public:
    bool enabled = false;
    uint32_t fps = 25;
    std::vector<Keyframe> keyframes;

    bool has_initial_keyframe() const { return !keyframes.empty() && keyframes[0].time == 0; }               // :79-88
    uint64_t get_frames_amount_to_render() const {                                                            // :90-98
        const uint64_t last = keyframes.empty() ? 0 : keyframes.back().time;
        return (uint64_t)std::floor((float)fps * ((float)last / 1000.0f));
    }
    bool has_animation() const {                                                                              // :100-103
        return enabled && get_frames_amount_to_render() > 0 && has_initial_keyframe() && keyframes.size() >= 2;
    }
    bool frame_exists(uint64_t frame) const { return has_animation() && frame < get_frames_amount_to_render(); } // scene.rs:1690-1693

    // :105-130: the keyframes around `frame` and the interpolation factor (NaN at the last keyframe: 1/0 * 0, as in Rust)
    void get_keyframes_for_frame(uint64_t frame, const Keyframe** first, const Keyframe** last, double* factor) const {
        const uint64_t timestamp = (uint64_t)std::floor((1000.0f / (float)fps) * (float)frame);
        *first = *last = &keyframes[0];
        for (size_t i = 0; i < keyframes.size(); i++)
            if (keyframes[i].time <= timestamp) { *first = &keyframes[i]; *last = (i + 1 >= keyframes.size()) ? &keyframes[i] : &keyframes[i + 1]; }
        const double pos = (double)(timestamp - (*first)->time), diff = (double)((*last)->time - (*first)->time);
        *factor = 1.0 / diff * pos;
    }

    // :132-205: interpolated T * Rz * Ry * Rx * S of one object for `frame` (column-major), false if it has no keyframes
    bool get_trans_for_frame(uint64_t frame, const std::string& object_name, float out[16]) const {
        const Keyframe *kf, *kl; double factor;
        get_keyframes_for_frame(frame, &kf, &kl, &factor);
        const Frame *a = nullptr, *b = nullptr;
        for (const Frame& f : kf->objects) if (f.object_name == object_name) { a = &f; break; }
        for (const Frame& f : kl->objects) if (f.object_name == object_name) { b = &f; break; }
        if (!a || !b) return false;
        const float f = (float)factor;
        auto lerp = [f](const std::optional<Vec3>& p, const std::optional<Vec3>& q, Vec3 d) {
            if (!p || !q) return d;
            return Vec3{p->x + f * (q->x - p->x), p->y + f * (q->y - p->y), p->z + f * (q->z - p->z)}; // helper::interpolate
        };
        const Vec3 t = lerp(a->translation, b->translation, Vec3{0.0f, 0.0f, 0.0f});
        const Vec3 sc = lerp(a->scale, b->scale, Vec3{1.0f, 1.0f, 1.0f});
        const Vec3 r = lerp(a->rotation, b->rotation, Vec3{0.0f, 0.0f, 0.0f});
        get_transformation(t, sc, r, out);
        return true;
    }

    // ShapeBasics::get_transformation on the identity (src/shape/mod.rs:708-729): T * Rz * Ry * Rx * S, f32, column-major
    static void get_transformation(Vec3 t, Vec3 s, Vec3 r, float out[16]) {
        float m[16]; ident(m);
        float f[16];
        ident(f); f[12] = t.x; f[13] = t.y; f[14] = t.z; mul(m, f);
        rot(2, r.z, f); mul(m, f);
        rot(1, r.y, f); mul(m, f);
        rot(0, r.x, f); mul(m, f);
        ident(f); f[0] = s.x; f[5] = s.y; f[10] = s.z; mul(m, f);
        std::memcpy(out, m, sizeof m);
    }

    // ShapeBasics::calc_inverse (src/shape/mod.rs:763-767) of an affine matrix; evaluated in double, rounded to f32
    static void inverse_affine(const float m[16], float out[16]) {
        const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        const double r[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                             (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                             (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det}; // row-major inverse of the 3x3
        const double tx = m[12], ty = m[13], tz = m[14];
        for (int k = 0; k < 16; k++) out[k] = 0.0f;
        for (int rr = 0; rr < 3; rr++) {
            for (int cc = 0; cc < 3; cc++) out[cc * 4 + rr] = (float)r[rr * 3 + cc];
            out[12 + rr] = (float)(-(r[rr * 3] * tx + r[rr * 3 + 1] * ty + r[rr * 3 + 2] * tz));
        }
        out[15] = 1.0f;
    }

    // Scene::apply_frame on the arrays rr_scene_update_transforms takes: `trans` / `trans_inv` hold n_items * 16 floats
    // (start from the items' own matrices); items named in the keyframes get their matrix REPLACED (apply_mat).
    // Returns false when the reference would not touch the scene.
    bool frame_transforms(const std::vector<std::string>& item_names, uint64_t frame, float* trans, float* trans_inv) const {
        if (!has_animation() || frame > get_frames_amount_to_render()) return false;
        for (size_t i = 0; i < item_names.size(); i++) {
            float m[16];
            if (get_trans_for_frame(frame, item_names[i], m)) std::memcpy(trans + 16 * i, m, sizeof m);
            inverse_affine(trans + 16 * i, trans_inv + 16 * i);
        }
        return true;
    }

private:
    static void ident(float* m) { for (int k = 0; k < 16; k++) m[k] = (k % 5 == 0) ? 1.0f : 0.0f; }
    static void rot(int axis, float a, float* m) {
        ident(m);
        const float c = (float)std::cos((double)a), s = (float)std::sin((double)a);
        if (axis == 0) { m[5] = c; m[9] = -s; m[6] = s; m[10] = c; }
        else if (axis == 1) { m[0] = c; m[8] = s; m[2] = -s; m[10] = c; }
        else { m[0] = c; m[4] = -s; m[1] = s; m[5] = c; }
    }
    static void mul(float* m, const float* f) { // m = m * f, f32 accumulation in nalgebra's column-axpy order
        float o[16];
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                float acc = m[r] * f[c * 4];
                for (int k = 1; k < 4; k++) acc = acc + m[k * 4 + r] * f[c * 4 + k];
                o[c * 4 + r] = acc;
            }
        std::memcpy(m, o, sizeof o);
    }
};

// A scene resident on one GPU (rr_scene), owned.
class DeviceScene {
public:
    DeviceScene(const rr_flat_scene& flat, int device = 0) {
        if (rr_scene_create(&flat, device, &h_) != RR_OK) { error_ = rr_last_error(); h_ = nullptr; }
    }
    ~DeviceScene() { if (h_) rr_scene_destroy(h_); }
    DeviceScene(const DeviceScene&) = delete;
    DeviceScene& operator=(const DeviceScene&) = delete;
    rr_scene* handle() const { return h_; }
    bool ok() const { return h_ != nullptr; }
    const std::string& error() const { return error_; }

private:
    rr_scene* h_ = nullptr;
    std::string error_;
};

// reference src/raytracing.rs:205-273: scene + config; the per-pixel `render(x, y)` is replaced by whole frames
class Raytracing {
public:
    std::shared_ptr<DeviceScene> scene;
    Camera camera; // Scene::cam in the reference (src/scene.rs:69-83)
    RaytracingConfig config;

    explicit Raytracing(std::shared_ptr<DeviceScene> s) : scene(std::move(s)) {}

    float gamma_encode(float linear) const { return std::pow(linear, 1.0f / 2.2f); } // src/raytracing.rs:231-235

    // Raytracing::pick (src/raytracing.rs:237-273): Some((id, distance)) of the item under the pixel, or None
    std::optional<std::pair<uint32_t, float>> pick(int x, int y) const {
        rr_pick_result r;
        const rr_camera c = camera.c_struct();
        if (rr_pick(scene->handle(), &c, x, y, &r) != RR_OK || !r.hit) return std::nullopt;
        return std::make_pair(r.object_id, r.distance);
    }
};

// reference src/renderer.rs:38-251
class RendererManager {
public:
    uint32_t thread_amount = 1; // one frame-level device call replaces the num_cpus - 2 workers (src/renderer.rs:67-71)
    uint32_t min_passes = 8;    // previews per frame (rr_render_progressive)

    RendererManager(int32_t width, int32_t height, std::shared_ptr<Raytracing> raytracing)
        : width_(width), height_(height), raytracing_(std::move(raytracing)) {}
    ~RendererManager() { stop(); join(); }

    void update_resolution(int32_t width, int32_t height) { width_ = width; height_ = height; } // :99-103

    // :105-172.  Returns at once; the frame renders on a worker thread.
    void start() {
        join();
        start_time_ = std::chrono::steady_clock::now();
        done_ms_ = 0;
        pixels_rendered_ = 0;
        cancel_ = 0;
        failed_ = false;
        running_ = true;
        const size_t n = (size_t)width_ * (size_t)height_;
        {
            std::lock_guard<std::mutex> lk(frame_mu_);
            image_.assign(n * 4, 0); normals_.assign(n * 3, 0.0f); depth_.assign(n, 0.0f); objects_.assign(n, 0u);
            fresh_ = false;
        }
        back_rgba_.assign(n * 4, 0); back_normal_.assign(n * 3, 0.0f); back_depth_.assign(n, 0.0f); back_ids_.assign(n, 0u);
        raytracing_->camera.init((uint32_t)width_, (uint32_t)height_);
        thread_ = std::thread([this]() { this->run(); });
    }

    // :174-198: ends the frame early; the buffers keep the last finished pass
    void stop() {
        if (!running_.exchange(false)) return;
        cancel_ = 1;
        join();
    }

    void restart(int32_t width, int32_t height) { stop(); update_resolution(width, height); start(); } // :201-208

    bool has_cells_left() const { return running_ && !is_done(); }                     // :210-213
    uint64_t get_rendered_pixels() const { return pixels_rendered_; }                  // :215-218
    bool is_running() const { return running_; }                                      // :220-226
    bool is_done() const { return pixels_rendered_ == (uint64_t)width_ * (uint64_t)height_; } // :228-231

    uint64_t check_and_get_elapsed_time() { // :233-246
        if (done_ms_ > 0) return done_ms_;
        const uint64_t ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start_time_).count();
        if (is_done()) done_ms_ = ms;
        return ms;
    }

    // Stands in for get_message_receiver (:248-251) + Run::apply_pixels (src/run.rs:506-545): calls `f` for every pixel
    // of the newest finished pass, once per pass.  Returns the number of pixels delivered.
    size_t drain(const std::function<void(const PixelData&)>& f) {
        std::lock_guard<std::mutex> lk(frame_mu_);
        if (!fresh_) return 0;
        fresh_ = false;
        const size_t n = (size_t)width_ * (size_t)height_;
        for (size_t i = 0; i < n; i++) {
            PixelData p;
            p.r = image_[4 * i]; p.g = image_[4 * i + 1]; p.b = image_[4 * i + 2];
            p.normal = Vec3{normals_[3 * i], normals_[3 * i + 1], normals_[3 * i + 2]};
            p.depth = depth_[i]; p.object_id = objects_[i];
            p.x = (int32_t)(i % (size_t)width_); p.y = (int32_t)(i / (size_t)width_);
            f(p);
        }
        return n;
    }

    // the four buffers Run::apply_pixels fills (src/run.rs:519-541); copies, safe while a frame renders
    void frame(std::vector<uint8_t>* rgba, std::vector<float>* normal = nullptr, std::vector<float>* depth = nullptr,
               std::vector<uint32_t>* object_id = nullptr) {
        std::lock_guard<std::mutex> lk(frame_mu_);
        if (rgba) *rgba = image_;
        if (normal) *normal = normals_;
        if (depth) *depth = depth_;
        if (object_id) *object_id = objects_;
    }

    // Run::post_processing (src/run.rs:588-600) -> run_post_processing (src/post_processing.rs:123-181) on the finished
    // frame: outline on object-id edges, then cavity shading from the normal buffer.  Returns the new image.
    std::vector<uint8_t> post_processing(bool cavity, bool outline, int device = 0) {
        std::vector<uint8_t> in; std::vector<float> nr; std::vector<uint32_t> ids;
        frame(&in, &nr, nullptr, &ids);
        std::vector<uint8_t> out(in.size());
        if (rr_post_process((uint32_t)width_, (uint32_t)height_, cavity ? 1 : 0, outline ? 1 : 0, in.data(), nr.data(), ids.data(), out.data(), device) != RR_OK) {
            std::lock_guard<std::mutex> lk(frame_mu_);
            error_ = rr_last_error();
            return in;
        }
        return out;
    }

    void wait() { join(); } // not in the reference: block until the frame is done or stopped
    bool failed() const { return failed_; }
    std::string last_error() { std::lock_guard<std::mutex> lk(frame_mu_); return error_; }
    uint32_t passes() const { return passes_; }

private:
    void join() { if (thread_.joinable()) thread_.join(); }

    static int on_pass(void* user, uint64_t done, uint64_t total) {
        RendererManager* m = (RendererManager*)user;
        m->publish();
        // the reference counts finished pixels; a pass finishes a share of every pixel's samples
        m->pixels_rendered_ = (uint64_t)m->width_ * (uint64_t)m->height_ * done / total;
        m->passes_++;
        return m->running_ ? 0 : 1;
    }

    void publish() {
        std::lock_guard<std::mutex> lk(frame_mu_);
        image_ = back_rgba_; normals_ = back_normal_; depth_ = back_depth_; objects_ = back_ids_;
        fresh_ = true;
    }

    void run() {
        passes_ = 0;
        const rr_camera cam = raytracing_->camera.c_struct();
        const rr_config cfg = raytracing_->config.c_struct();
        rr_frame out{back_rgba_.data(), back_normal_.data(), back_depth_.data(), back_ids_.data()};
        const int rc = rr_render_progressive(raytracing_->scene->handle(), &cam, &cfg, nullptr, &out, min_passes, &RendererManager::on_pass, this, &cancel_);
        if (rc == RR_OK) {
            publish();
            pixels_rendered_ = (uint64_t)width_ * (uint64_t)height_;
        } else if (rc != RR_ERR_CANCELLED) {
            std::lock_guard<std::mutex> lk(frame_mu_);
            error_ = rr_last_error();
            failed_ = true;
        }
        running_ = false;
    }

    int32_t width_, height_;
    std::shared_ptr<Raytracing> raytracing_;
    std::thread thread_;
    std::atomic<bool> running_{false}, failed_{false};
    std::atomic<uint64_t> pixels_rendered_{0};
    std::atomic<uint32_t> passes_{0};
    volatile int cancel_ = 0;
    std::chrono::steady_clock::time_point start_time_ = std::chrono::steady_clock::now();
    uint64_t done_ms_ = 0;
    std::mutex frame_mu_;
    std::vector<uint8_t> image_, back_rgba_;
    std::vector<float> normals_, depth_, back_normal_, back_depth_;
    std::vector<uint32_t> objects_, back_ids_;
    bool fresh_ = false;
    std::string error_;
};

} // namespace rustray
