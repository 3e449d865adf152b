// rr_api.hip — host side of librustray_hip.so: the C ABI of include/rustray_hip.h.
//
// Replaces the frame loop of `RendererManager::start` (reference
// src/renderer.rs:105-172): instead of a shuffled queue of 2x2-pixel cells and
// num_cpus-2 worker threads calling Raytracing::render per pixel, one call
// uploads the frame constants and drives the wavefront kernels of
// rr_kernels.hip over batches of primary samples.
#include "../../include/rustray_hip.h"
#include "rr_bvh.h"
#include "rr_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <limits>
#include <map>
#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rr_kernels.hip"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string tl_error;

static int fail(int code, const char* fmt, ...) noexcept {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try { tl_error = buf; } catch (...) { /* no memory for the message: the code still says what happened */ }
    return code;
}

// ---------------------------------------------------------------------------
// Nothing unwinds across the C ABI (include/rustray_hip.h: "no function aborts or throws").  The host is Rust built with
// panic = "abort" (reference Cargo.toml:9-11); a C++ exception that reached one of its frames would be undefined behaviour.
// Every extern "C" entry point below is a function-try-block that ends in RR_GUARD_END: std::bad_alloc (the std::vectors
// sized by the caller's scene) becomes RR_ERR_OUT_OF_MEMORY, anything else RR_ERR_DEVICE with what() in rr_last_error().
// Host worker threads (mesh tree builds, one thread per device in rr_render_multi) run under `Workers`: an exception inside
// a worker is carried to the calling thread and rethrown there, a thread that cannot be started is not fatal (the caller
// does that work itself), and the destructor joins -- no path ends in std::terminate.
// ---------------------------------------------------------------------------
static int guard_fail(const char* fn) noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(RR_ERR_OUT_OF_MEMORY, "%s: out of host memory", fn); }
    catch (const std::exception& e) { return fail(RR_ERR_DEVICE, "%s: %s", fn, e.what()); }
    catch (...) { return fail(RR_ERR_DEVICE, "%s: unknown exception", fn); }
}
#define RR_GUARD_END(fn) catch (...) { return guard_fail(fn); }

struct Workers {
    std::vector<std::thread> threads;
    std::atomic_flag taken = ATOMIC_FLAG_INIT;
    std::exception_ptr first; // written by the one worker that wins `taken`, read after join()
    Workers() = default;
    Workers(const Workers&) = delete;
    Workers& operator=(const Workers&) = delete;
    ~Workers() { join(); }
    // runs f() on a new thread; false = no thread could be started (the caller runs f itself)
    template <class F> bool spawn(F f) {
        try {
            threads.emplace_back([this, f]() mutable { run(f); });
            return true;
        } catch (...) { return false; }
    }
    // f() on the calling thread, under the same net
    template <class F> void run(F& f) noexcept {
        try { f(); }
        catch (...) { if (!taken.test_and_set()) first = std::current_exception(); }
    }
    void join() noexcept { for (std::thread& t : threads) if (t.joinable()) t.join(); }
    void join_and_rethrow() { join(); if (first) std::rethrow_exception(first); }
};

// Test-only fault injection (tests/test_abi.py, tests/test_gpu_guard.py): rr_test_fault("point", kind, skip) arms ONE fault; the
// (skip + 1)-th crossing of RR_FAULT_POINT("point") on any thread throws std::bad_alloc (kind 1), std::runtime_error (2) or
// an int (3) and disarms.  Not armed (always, outside the tests): one relaxed atomic load per crossing, and the points sit
// outside every per-ray and per-triangle loop.
static std::atomic<int> g_fault_kind{0};
static std::atomic<int> g_fault_skip{0};
static char g_fault_point[64] = "";
static void fault_point(const char* name) {
    if (g_fault_kind.load(std::memory_order_relaxed) == 0 || strcmp(name, g_fault_point) != 0) return;
    if (g_fault_skip.fetch_sub(1) > 0) return;
    const int kind = g_fault_kind.exchange(0);
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::runtime_error(std::string("injected fault at ") + name);
    if (kind == 3) throw 42;
}
#define RR_FAULT_POINT(name) fault_point(name)
extern "C" int rr_test_fault(const char* point, int kind, int skip) {
    g_fault_kind.store(0);
    if (!point || kind < 0 || kind > 3 || strlen(point) >= sizeof g_fault_point) return fail(RR_ERR_INVALID_ARGUMENT, "rr_test_fault: bad arguments");
    strcpy(g_fault_point, point);
    g_fault_skip.store(skip < 0 ? 0 : skip);
    g_fault_kind.store(kind);
    return RR_OK;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? RR_ERR_OUT_OF_MEMORY : RR_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------
// device buffer helper
// ---------------------------------------------------------------------------
// Owns one device allocation (freed on destruction: every exit path of rr_scene_create and the scratch buffers of
// rr_pick / rr_post_process release what they hold).  Move-only.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
    ~DevBuf() { release(); }
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct TimedLaunch { hipEvent_t a, b; int kind; }; // kind: 0 closest-hit (deeper levels), 4 (level 1); 1 shadow, 6 (level 1); 2 shade, 5 (level 1); 3 binning
struct ItemHost { uint32_t kind; int32_t material, material_cache; bool visible, flip_normals, mesh_has_normals, mesh_degenerate; int32_t mesh; };

struct rr_scene {
    int device = 0;
    int n_cus = 256;
    std::mutex mu;
    // scene data
    DevBuf items, nodes4, tnodes4, tris, trix, attrs, face_slot, materials, textures, texels, lights, flat_normals;
    DSceneView view{};
    std::vector<DItem> h_items;
    std::vector<uint32_t> h_slot_face; // per mesh triangle: leaf-order slot -> original face index (rr_trace_rays reports the reference's face id)
    std::vector<ItemHost> item_host; // what rr_scene_update_materials needs to rebuild the item flag words
    // per item: the extent of its surface along the rows of its transform (k_item_spans: minima, maxima, largest |local coordinate|; 9 doubles),
    // read back after every upload of the items' transforms; the top level's surface boxes are derived from it (exact_world_box)
    std::vector<double> h_spans;
    DevBuf spans, item_chunks;              // item_chunks: (item, first triangle) per workgroup of k_world_normals / k_item_spans (RR_ITEM_CHUNK triangles each)
    std::vector<uint32_t> h_chunk_item;     // the item of every chunk (chunks of one item are consecutive)
    std::vector<uint32_t> tex_width;
    std::vector<DTexture> h_textures; // descriptors of the uploaded images (copied into the material records, make_dmaterial)
    uint32_t n_materials = 0;
    uint32_t n_enabled_lights = 0;
    uint32_t tlas_node_capacity = 0;
    std::vector<float4> h_item_boxes; // padded world boxes per item (lo, hi), filled by build_tlas
    DevBuf item_boxes;
    DevBuf sq_valid; // one 64-bit word per (enabled light, 64 shadow slots): which lanes hold a ray
    double tlas_reach[3] = {0.0, 0.0, 0.0}; // the top level's boxes are padded for ray origins within +-tlas_reach (build_tlas)
    double tlas_floor[3] = {0.0, 0.0, 0.0}; // ... and never for less than this: the items' own extent
    int tlas_depth_limit = RR_TLAS_MAX_DEPTH, blas_depth_limit = RR_BLAS_MAX_DEPTH; // shares of the traversal stack, see rr_scene_create
    // frame state (grown on demand, reused across frames)
    DevBuf hit1;      // hit records of depth level 1 (the primary rays are derived from their index, not stored)
    DevBuf arena[4];  // ray records of the deeper live depth levels, SoA: r0 r1 r2 hit
    size_t arena_cap = 0; // rays
    uint32_t arena_factor = 2; // arena rays per primary ray of a batch; doubled after a frame that had to slice levels
    DevBuf sq[3];
    size_t sq_cap = 0;
    DevBuf acc_rgb, acc_normal, acc_depth, acc_id, acc_flags, shade_const;
    DevBuf region_xy, trace_order, sample_xy, pool, counters; // region_xy: pixel of each accumulator slot; trace_order: its output index
    std::vector<DevBuf> pool_more; // further segments of per-batch counters, for batches with very many launches (kept for the next frame)
    DevBuf tmp_out[4];
    DevBuf multi_part[4], multi_cat[4]; // rr_render_multi: this device's compact buffers; on device slot 0 the concatenation of all
    hipStream_t multi_stream = nullptr; // rr_render_multi: this handle's own non-blocking stream (created on first use)
    void* multi_stage[4] = {nullptr, nullptr, nullptr, nullptr}; size_t multi_stage_bytes[4] = {0, 0, 0, 0}; // pinned staging, devices without peer access
    std::vector<uint32_t> h_region_xy;
    rr_region region_cached{0, 0, 0, 0};
    uint32_t region_w = 0, region_h = 0;
    // stats
    rr_frame_stats stats{};
    bool stats_final = false; // s->stats already holds the sums over the passes of rr_render_progressive_tiles (nothing to collect from the device)
    bool profiling = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t frame_a = nullptr, frame_b = nullptr, count_ready = nullptr;
    hipStream_t last_stream = nullptr; // frame state (queues, accumulators) is shared: frames on different streams are serialised
    uint32_t* h_count = nullptr; // pinned: level sizes read back between depth levels
    rr_tuning tuning{};          // rr_scene_set_tuning; all zero = automatic
    std::vector<uint16_t> table_cache; uint16_t table_samples = 0; // built-in sub-sample table of the last sample count
    // every device buffer is a DevBuf member (freed by its destructor, on the scene's device)
    ~rr_scene() {
        (void)hipSetDevice(device);
        for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
        for (auto& t : timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        if (frame_a) (void)hipEventDestroy(frame_a);
        if (frame_b) (void)hipEventDestroy(frame_b);
        if (count_ready) (void)hipEventDestroy(count_ready);
        if (h_count) (void)hipHostFree(h_count);
        if (multi_stream) (void)hipStreamDestroy(multi_stream);
        for (void* p : multi_stage) if (p) (void)hipHostFree(p);
    }
};

static const uint32_t POOL_WORDS = 1u << 22; // per-batch counters (level sizes, fetch heads, shadow shard counts): 16 MB, zeroed per batch

// ---------------------------------------------------------------------------
// the reference's sub-sample table: StdRng::seed_from_u64(0) + shuffle + truncate
// (reference src/raytracing.rs:290-313; rand 0.8: ChaCha12 core, PCG32 seed
// expansion, Fisher-Yates from the back with widening-multiply rejection)
// ---------------------------------------------------------------------------
namespace {

inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

struct ChaCha12 {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t block[16];
    int pos = 16;
    explicit ChaCha12(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 8; i++) { // SeedableRng::seed_from_u64
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key[i] = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
    }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                           key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t x[16];
        memcpy(x, in, sizeof x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; r++) { // 12 rounds = 6 double rounds
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) block[i] = x[i] + in[i];
        counter++;
        pos = 0;
    }
    uint32_t next_u32() { if (pos >= 16) refill(); return block[pos++]; }
    uint32_t below(uint32_t range) { // UniformInt<u32>::sample_single(0, range)
        uint32_t zone = (range << __builtin_clz(range)) - 1u;
        for (;;) {
            uint64_t m = (uint64_t)next_u32() * range;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
};

uint32_t cell_size_of(uint16_t samples) {
    if (samples <= 1) return 1;
    uint16_t v = (uint16_t)(samples + 2);
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p / 2;
}

} // namespace

extern "C" int rr_sample_table(uint16_t samples, uint16_t* xy_out, uint32_t* cell_size_out) try {
    if (!xy_out && samples) return fail(RR_ERR_INVALID_ARGUMENT, "rr_sample_table: xy_out is NULL");
    if (samples > RR_MAX_SAMPLES) return fail(RR_ERR_UNSUPPORTED, "samples %u > %u", (unsigned)samples, RR_MAX_SAMPLES);
    uint32_t cs = cell_size_of(samples);
    std::vector<uint32_t> cells;
    try { cells.resize((size_t)cs * cs); } // 268 MB at the largest cell size: a failure must not cross the C ABI as an exception
    catch (const std::exception&) { return fail(RR_ERR_OUT_OF_MEMORY, "rr_sample_table: no host memory for %u x %u cells", cs, cs); }
    size_t k = 0;
    for (uint32_t xi = 0; xi < cs; xi++)
        for (uint32_t yi = 0; yi < cs; yi++) cells[k++] = xi | (yi << 16);
    ChaCha12 rng(0);
    for (size_t i = cells.size(); i-- > 1;) std::swap(cells[i], cells[rng.below((uint32_t)(i + 1))]);
    for (uint32_t s = 0; s < samples && s < cells.size(); s++) {
        xy_out[2 * s] = (uint16_t)(cells[s] & 0xffffu);
        xy_out[2 * s + 1] = (uint16_t)(cells[s] >> 16);
    }
    if (cell_size_out) *cell_size_out = cs;
    return RR_OK;
} RR_GUARD_END("rr_sample_table")

// ---------------------------------------------------------------------------
// misc entry points
// ---------------------------------------------------------------------------
extern "C" uint32_t rr_abi_version(void) { return RR_ABI_VERSION; }
extern "C" int rr_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} RR_GUARD_END("rr_device_count")
extern "C" const char* rr_last_error(void) { return tl_error.c_str(); }

// xy: the region's pixels in OUTPUT order (tile order, row-major inside the tile; the ABI contract).
// trace_order (optional): a permutation of region indices = the order of the ACCUMULATOR SLOTS, in which
// primary rays are generated: 8x8-pixel blocks inside each tile, so that the 64 lanes of a wave start as one
// compact bundle of rays whatever the tile shape is (32x8 tiles traced in row-major order cost 4 % more than
// 8x8 blocks on sponza_syn) and add to 64 consecutive accumulator words.
static void fill_region(uint32_t w, uint32_t h, const rr_region& rg, std::vector<uint32_t>* xy, std::vector<uint32_t>* trace_order = nullptr) {
    xy->clear();
    if (trace_order) trace_order->clear();
    uint32_t tx = (w + rg.tile_w - 1) / rg.tile_w, ty = (h + rg.tile_h - 1) / rg.tile_h;
    for (uint32_t t = rg.rank; t < tx * ty; t += rg.n_ranks) {
        uint32_t x0 = (t % tx) * rg.tile_w, y0 = (t / tx) * rg.tile_h;
        uint32_t x1 = std::min(x0 + rg.tile_w, w), y1 = std::min(y0 + rg.tile_h, h);
        const uint32_t base = (uint32_t)xy->size(), tw = x1 - x0;
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) xy->push_back(x | (y << 16));
        if (trace_order)
            for (uint32_t by = y0; by < y1; by += 8)
                for (uint32_t bx = x0; bx < x1; bx += 8)
                    for (uint32_t y = by; y < std::min(by + 8, y1); y++)
                        for (uint32_t x = bx; x < std::min(bx + 8, x1); x++) trace_order->push_back(base + (y - y0) * tw + (x - x0));
    }
}
static int check_region(uint32_t w, uint32_t h, const rr_region* rg) {
    if (!rg) return fail(RR_ERR_INVALID_ARGUMENT, "region is NULL");
    if (rg->tile_w == 0 || rg->tile_h == 0 || rg->n_ranks == 0 || rg->rank >= rg->n_ranks)
        return fail(RR_ERR_INVALID_ARGUMENT, "bad region: tile %ux%u rank %u of %u", rg->tile_w, rg->tile_h, rg->rank, rg->n_ranks);
    if (w == 0 || h == 0 || w > 65535u || h > 65535u) return fail(RR_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", w, h);
    return RR_OK;
}
extern "C" uint64_t rr_region_pixel_count(uint32_t width, uint32_t height, const rr_region* rg) {
    if (check_region(width, height, rg) != RR_OK) return 0;
    uint32_t tx = (width + rg->tile_w - 1) / rg->tile_w, ty = (height + rg->tile_h - 1) / rg->tile_h;
    uint64_t n = 0;
    for (uint32_t t = rg->rank; t < tx * ty; t += rg->n_ranks) {
        uint32_t x0 = (t % tx) * rg->tile_w, y0 = (t / tx) * rg->tile_h;
        n += (uint64_t)(std::min(x0 + rg->tile_w, width) - x0) * (std::min(y0 + rg->tile_h, height) - y0);
    }
    return n;
}

// ---------------------------------------------------------------------------
// scene validation + upload
// ---------------------------------------------------------------------------
static bool finite16(const float* m) { for (int i = 0; i < 16; i++) if (!std::isfinite(m[i])) return false; return true; }

static int validate_scene(const rr_flat_scene* fs) {
    if (!fs) return fail(RR_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (fs->abi_version != RR_ABI_VERSION) return fail(RR_ERR_INVALID_ARGUMENT, "abi_version %u, library speaks %u", fs->abi_version, RR_ABI_VERSION);
    if ((fs->n_items && !fs->items) || (fs->n_meshes && !fs->meshes) || (fs->n_materials && !fs->materials) ||
        (fs->n_textures && !fs->textures) || (fs->n_lights && !fs->lights))
        return fail(RR_ERR_INVALID_ARGUMENT, "array pointer is NULL with a non-zero count");
    if (fs->n_items >= (1u << 27)) return fail(RR_ERR_UNSUPPORTED, "%u items (the shadow-ray record keeps the item index in 27 bits)", fs->n_items);
    for (uint32_t i = 0; i < fs->n_textures; i++) {
        const rr_texture& t = fs->textures[i];
        if ((uint64_t)t.width * t.height > 0 && !t.rgba8) return fail(RR_ERR_INVALID_ARGUMENT, "texture %u has no pixels", i);
        if (t.width > 32768u || t.height > 32768u) return fail(RR_ERR_UNSUPPORTED, "texture %u is %ux%u", i, t.width, t.height);
    }
    for (uint32_t i = 0; i < fs->n_materials; i++)
        for (int k = 0; k < RR_TEX_COUNT; k++) {
            int32_t t = fs->materials[i].texture[k];
            if (t >= (int32_t)fs->n_textures) return fail(RR_ERR_INVALID_ARGUMENT, "material %u texture slot %d = %d out of range", i, k, t);
        }
    for (uint32_t i = 0; i < fs->n_meshes; i++) {
        const rr_mesh& m = fs->meshes[i];
        if ((m.n_vertices && !m.positions) || (m.n_triangles && !m.indices)) return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: missing positions/indices", i);
        if ((m.n_uvs && !m.uvs) || (m.n_uv_faces && !m.uv_indices) || (m.n_normals && !m.normals) || (m.n_normal_faces && !m.normal_indices))
            return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: attribute pointer is NULL with a non-zero count", i);
        if (m.n_triangles >= (1u << 28)) return fail(RR_ERR_UNSUPPORTED, "mesh %u: %u triangles", i, m.n_triangles);
        for (size_t k = 0; k < (size_t)m.n_triangles * 3; k++)
            if (m.indices[k] >= m.n_vertices) return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: vertex index %u out of range", i, m.indices[k]);
        for (size_t k = 0; k < (size_t)m.n_uv_faces * 3; k++)
            if (m.uv_indices[k] >= m.n_uvs) return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: uv index %u out of range", i, m.uv_indices[k]);
        for (size_t k = 0; k < (size_t)m.n_normal_faces * 3; k++)
            if (m.normal_indices[k] >= m.n_normals) return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: normal index %u out of range", i, m.normal_indices[k]);
        // Mesh::get_normal indexes normals_indices[face] unchecked (reference src/shape/mesh.rs:216): the reference would panic
        if (m.n_normals > 0 && m.n_normal_faces > 0 && m.n_normal_faces < m.n_triangles)
            return fail(RR_ERR_INVALID_ARGUMENT, "mesh %u: %u normal faces for %u triangles (the reference panics on this)", i, m.n_normal_faces, m.n_triangles);
    }
    for (uint32_t i = 0; i < fs->n_items; i++) {
        const rr_item& it = fs->items[i];
        if (it.kind != RR_ITEM_SPHERE && it.kind != RR_ITEM_MESH) return fail(RR_ERR_INVALID_ARGUMENT, "item %u: kind %u", i, it.kind);
        if (it.material < 0 || it.material >= (int32_t)fs->n_materials || it.material_cache < 0 || it.material_cache >= (int32_t)fs->n_materials)
            return fail(RR_ERR_INVALID_ARGUMENT, "item %u: material index out of range", i);
        for (int k = 0; k < RR_TEX_COUNT; k++)
            if (fs->materials[it.material_cache].texture[k] >= 0)
                return fail(RR_ERR_INVALID_ARGUMENT, "item %u: material_cache must not carry textures (reference src/shape/mod.rs:769-772)", i);
        if (it.kind == RR_ITEM_MESH && (it.mesh < 0 || it.mesh >= (int32_t)fs->n_meshes)) return fail(RR_ERR_INVALID_ARGUMENT, "item %u: mesh index %d", i, it.mesh);
        if (!finite16(it.trans) || !finite16(it.trans_inv)) return fail(RR_ERR_INVALID_ARGUMENT, "item %u: non-finite transform", i);
    }
    for (uint32_t i = 0; i < fs->n_lights; i++)
        if (fs->lights[i].light_type > RR_LIGHT_SPOT) return fail(RR_ERR_INVALID_ARGUMENT, "light %u: type %u", i, fs->lights[i].light_type);
    return RR_OK;
}

// Material (reference src/shape/mod.rs:95-134) -> device record; has_texture = the slot names an image of width > 0
static DMaterial make_dmaterial(const rr_material& m, const std::vector<uint32_t>& tex_width, const std::vector<DTexture>& dtex) {
    DMaterial d;
    memset(&d, 0, sizeof d);
    for (int k = 0; k < 3; k++) { d.ambient[k] = m.ambient_color[k]; d.base[k] = m.base_color[k]; d.specular[k] = m.specular_color[k]; }
    d.alpha = m.alpha; d.shininess = m.shininess; d.reflectivity = m.reflectivity; d.refraction_index = m.refraction_index;
    d.normal_map_strength = m.normal_map_strength; d.shadow_softness = m.shadow_softness; d.roughness = m.roughness;
    d.cos_shadow_softness = rr_cos(m.shadow_softness * RR_PI_F); d.cos_roughness = rr_cos(m.roughness * RR_PI_F); // jitter()'s z_lo, see DMaterial
    bool any = false;
    uint32_t slots = 0u;
    for (int k = 0; k < RR_TEX_COUNT; k++) {
        d.tex[k] = m.texture[k];
        if (m.texture[k] >= 0) { const DTexture& t = dtex[m.texture[k]]; d.texd[k].offset = t.offset; d.texd[k].width = t.width; d.texd[k].height = t.height; }
        if (m.texture[k] >= 0 && tex_width[m.texture[k]] > 0) { any = true; slots |= RR_MF_TEX_SLOT0 << k; }
    }
    d.flags = slots | (m.texture_filtering_nearest ? RR_MF_NEAREST : 0u) | (m.receive_shadow ? RR_MF_RECEIVE_SHADOW : 0u) |
              (m.monte_carlo ? RR_MF_MONTE_CARLO : 0u) | (any ? RR_MF_ANY_TEX : 0u);
    return d;
}

// The flag word of an item: what Raytracing::trace reads of the texture-less material cache (src/raytracing.rs:450-458),
// intersect_b_box's `solid` (src/shape/mesh.rs:51-59) and the occluder-alpha-map test of the shadow code (:899-912).
static uint32_t item_flags(const ItemHost& it, const rr_material& cache, const rr_material& full, const std::vector<uint32_t>& tex_width) {
    uint32_t f = 0;
    if (it.visible) f |= RR_IF_VISIBLE;
    if (it.flip_normals) f |= RR_IF_FLIP_NORMALS;
    if (cache.alpha > 0.0f) f |= RR_IF_CACHE_ALPHA_POS;
    if (cache.cast_shadow) f |= RR_IF_CACHE_CAST_SHADOW;
    if (cache.reflection_only) f |= RR_IF_CACHE_REFL_ONLY;
    if (!(cache.alpha < 1.0f) && cache.backface_cullig) f |= RR_IF_SOLID_BASE; // the cache never has textures
    if (full.texture[RR_TEX_ALPHA] >= 0 && tex_width[full.texture[RR_TEX_ALPHA]] > 0) f |= RR_IF_OCCLUDER_ALPHA_TEX;
    if (it.kind == RR_ITEM_SPHERE) f |= RR_IF_SPHERE | RR_IF_UV_MAY_BE_NAN;
    else {
        if (cache.smooth_shading && it.mesh_has_normals) f |= RR_IF_SMOOTH;
        if (it.mesh_degenerate) f |= RR_IF_UV_MAY_BE_NAN;
    }
    return f;
}

static void fill_item_matrices(DItem& d, const float* trans, const float* inv) {
    // rows of the column-major matrices
    d.inv0 = make_float4(inv[0], inv[4], inv[8], inv[12]);
    d.inv1 = make_float4(inv[1], inv[5], inv[9], inv[13]);
    d.inv2 = make_float4(inv[2], inv[6], inv[10], inv[14]);
    d.inv3 = make_float4(inv[3], inv[7], inv[11], inv[15]);
    d.tr0 = make_float4(trans[0], trans[4], trans[8], trans[12]);
    d.tr1 = make_float4(trans[1], trans[5], trans[9], trans[13]);
    d.tr2 = make_float4(trans[2], trans[6], trans[10], trans[14]);
}

// ---------------------------------------------------------------------------
// top level: world boxes over the items
// ---------------------------------------------------------------------------
// The reference has no world-space test: a ray is moved into an item's space with the item's f32 inverse matrix and
// tested there against the local box (Shape::intersect_b_box, src/shape/mod.rs:80-105).  The top-level tree may only
// skip an item if that local test would fail, so an item's world box is the exact box of its transformed local corners
// (Bounded::aabb, src/shape/mod.rs:48-78; in double) grown by a bound on how far the f32 local ray can sit from the
// true one, mapped back to world space.  With M, N the given matrix and inverse, t, t' their translations, u = 2^-24,
// and origins with |o_c| <= reach_c:  the local ray is  N o + t' + do,  N d + dd  with  |do| <= g (|N| |o| + |t'|),
// |dd| <= g |N| |d|  (g = 16 u covers the four-term dot products and the local slab test's own rounding), so a point of
// it maps to the world point  (o + s d) + [E o + e + M do] + s [E d + M dd],  E = M N - I,  e = M t' + t  (N is an f32
// inverse: E is of the order u cond(M)).  s |d| is at most reach + the box's own extent, which bounds the last term.
// Far from the origin, or with a badly conditioned transform, this is orders of magnitude more than float spacing
// (tools/fuzz_parity.py far: a sheared sphere 1e5 away needs 15 units on a 10-unit box); on ordinary scenes it is
// ~1e-6 of the scene size.  An inverse with a projective bottom row gets an unbounded box (always a candidate).
//
// Tighter than the corners, for meshes: a mesh only ever contributes through a triangle its own tree lets the ray reach (a candidate
// whose box is hit and whose triangles are missed changes nothing: src/raytracing.rs:471-487 looks at `intersect`'s result only),
// and the walks of rr_kernels.hip test a triangle only under a leaf box that the local ray passes -- boxes of the mesh's own
// vertices, padded by 4e-6 of their coordinates (rr_bvh.cpp).  So whatever the triangle test then reports, NaN included, it
// reports for a ray that passes the padded box of the mesh's VERTICES; in world space that is the box of the transformed
// vertices, which for a rotated item is much smaller than the box of the rotated local box (a unit cube turned by 45 degrees
// about two axes: 1.7 x per axis).  The world box is the intersection of the two, grown by the leaf padding mapped to world
// space; the padding of padded_world_box (how far the f32 local ray sits from the true one) applies as before.
// NOT for balls: ray_ball has no box in front of it, and where its arithmetic overflows (a tiny ball: local coordinates
// ~1e12) it answers Some(NaN) for ANY ray that passes the local box -- tests/golden/fuzz_568 holds such a scene -- so a ball
// keeps the box of its local box's corners.  Not for a mesh whose tree is a single leaf either (nothing is culled in front of
// its triangles).  The extent of the vertices along the transform's rows comes from the device (k_item_spans), where the triangles live.
struct WorldBox { double lo[3], hi[3]; bool tight[3]; double ext[3]; }; // tight[r]: axis r comes from the vertices; ext: largest |local coordinate| per local axis
// `span`: the 9 doubles k_item_spans wrote for this item (extent of the mesh's vertices along the transform's rows), or NULL = corners only
static WorldBox exact_world_box(const DItem& it, const double* span) {
    WorldBox b;
    const float4 rows[3] = {it.tr0, it.tr1, it.tr2};
    for (int r = 0; r < 3; r++) { b.lo[r] = 1e300; b.hi[r] = -1e300; b.tight[r] = false; b.ext[r] = 0.0; }
    for (int c = 0; c < 8; c++) {
        const double p[3] = {(c & 1) ? it.bmax[0] : it.bmin[0], (c & 2) ? it.bmax[1] : it.bmin[1], (c & 4) ? it.bmax[2] : it.bmin[2]};
        for (int r = 0; r < 3; r++) {
            const double v = (double)rows[r].x * p[0] + (double)rows[r].y * p[1] + (double)rows[r].z * p[2] + (double)rows[r].w;
            b.lo[r] = std::min(b.lo[r], v); b.hi[r] = std::max(b.hi[r], v);
        }
    }
    if (!(it.flags & RR_IF_SPHERE) && it.root4 >= 0 && span) {
        bool finite = true;
        for (int k = 0; k < 9; k++) finite = finite && std::isfinite(span[k]);
        if (finite) {
            for (int c = 0; c < 3; c++) b.ext[c] = span[6 + c]; // what the leaf padding is relative to
            for (int r = 0; r < 3; r++) {
                const double mx = rows[r].x, my = rows[r].y, mz = rows[r].z;
                const double leaf_pad = 2.0e-5 * (std::fabs(mx) * b.ext[0] + std::fabs(my) * b.ext[1] + std::fabs(mz) * b.ext[2]) + 1e-30;
                const double tlo = span[r] + (double)rows[r].w - leaf_pad, thi = span[3 + r] + (double)rows[r].w + leaf_pad;
                if (std::isfinite(tlo) && std::isfinite(thi) && tlo <= thi && (tlo > b.lo[r] || thi < b.hi[r])) { b.lo[r] = std::max(b.lo[r], tlo); b.hi[r] = std::min(b.hi[r], thi); b.tight[r] = true; }
            }
        }
    }
    return b;
}
static void padded_world_box(const DItem& it, const WorldBox& b, const double reach[3], float* lo, float* hi) {
    const double M[3][4] = {{it.tr0.x, it.tr0.y, it.tr0.z, it.tr0.w}, {it.tr1.x, it.tr1.y, it.tr1.z, it.tr1.w}, {it.tr2.x, it.tr2.y, it.tr2.z, it.tr2.w}};
    const double N[3][4] = {{it.inv0.x, it.inv0.y, it.inv0.z, it.inv0.w}, {it.inv1.x, it.inv1.y, it.inv1.z, it.inv1.w}, {it.inv2.x, it.inv2.y, it.inv2.z, it.inv2.w}};
    const bool affine = it.inv3.x == 0.0f && it.inv3.y == 0.0f && it.inv3.z == 0.0f && it.inv3.w == 1.0f;
    const double g = 16.0 / 16777216.0;
    for (int r = 0; r < 3; r++) {
        double pad = 0.0;
        for (int c = 0; c < 3; c++) {
            double e = (r == c) ? -1.0 : 0.0, a = 0.0;
            for (int k = 0; k < 3; k++) { e += M[r][k] * N[k][c]; a += std::fabs(M[r][k]) * std::fabs(N[k][c]); }
            const double extent = std::max(std::fabs(b.lo[c]), std::fabs(b.hi[c]));
            pad += (std::fabs(e) + g * a) * (2.0 * reach[c] + extent);
        }
        double et = M[r][3], at = 0.0;
        for (int k = 0; k < 3; k++) { et += M[r][k] * N[k][3]; at += std::fabs(M[r][k]) * std::fabs(N[k][3]); }
        pad += std::fabs(et) + g * at;
        if (b.tight[r]) {
            // a leaf's slab test lets a ray through whose entry and exit distances differ by up to 8e-6 of themselves (RR_CHILD): planes moved
            // by that share of the way travelled along a local axis, which is at most the local reach plus the mesh's own extent
            for (int c = 0; c < 3; c++) {
                double way = std::fabs(N[c][3]) + b.ext[c];
                for (int k = 0; k < 3; k++) way += std::fabs(N[c][k]) * 2.0 * reach[k];
                pad += std::fabs(M[r][c]) * 1.0e-5 * way;
            }
        }
        pad = 2.0 * pad + 1e-6 * std::max(std::fabs(b.lo[r]), std::fabs(b.hi[r])) + 1e-30; // + float rounding of the box and of the walk's plane distances
        lo[r] = (float)(b.lo[r] - pad); hi[r] = (float)(b.hi[r] + pad);
        if (!affine || !std::isfinite(lo[r]) || lo[r] < -3.0e38f) lo[r] = -3.0e38f;
        if (!affine || !std::isfinite(hi[r]) || hi[r] > 3.0e38f) hi[r] = 3.0e38f;
    }
}

// Builds the top-level tree over s->h_items for ray origins within +-reach (grown to cover the items themselves: the
// origins of secondary and shadow rays lie on them).
struct TlasTrees { std::vector<DNode4> corner, surface; int32_t root = (int32_t)0x80000000, root_surface = (int32_t)0x80000000; bool has_surface = false; };
static int tlas_tree(rr_scene* s, const float* lo, const float* hi, uint32_t n, std::vector<DNode4>* tlas4, int32_t* root4);
static int build_tlas(rr_scene* s, const double want_reach[3], TlasTrees* trees) {
    uint32_t n = (uint32_t)s->h_items.size();
    *trees = TlasTrees();
    for (int c = 0; c < 3; c++) s->tlas_reach[c] = want_reach[c];
    if (n == 0) return RR_OK; // empty scene: every walk ends at once (both roots RR_SENTINEL)
    // two boxes per item: the box of its local box's corners -- what the tree is built over and what shadow packets are tested against:
    // the shadow query orders items by the distance at which the LOCAL box is entered, and prunes by it, which only a world box that
    // contains the local box bounds from below -- and the box of its surface (exact_world_box), which the closest-hit packets use:
    // there an item matters through its nearest hit alone, and that lies in the tighter box
    std::vector<WorldBox> exact(n), surf(n);
    for (uint32_t i = 0; i < n; i++) {
        exact[i] = exact_world_box(s->h_items[i], nullptr);
        surf[i] = exact_world_box(s->h_items[i], s->h_spans.size() == 9 * (size_t)n ? &s->h_spans[9 * (size_t)i] : nullptr);
        for (int c = 0; c < 3; c++) {
            const double m = std::max(std::fabs(exact[i].lo[c]), std::fabs(exact[i].hi[c])) * 1.001 + 0.01; // + the shadow bias along the normal
            if (std::isfinite(m)) s->tlas_reach[c] = std::max(s->tlas_reach[c], m);
        }
    }
    // Can some ball's ray_toi_with_ball overflow (b * b, a * c beyond f32: delta = NaN and the ball answers Some(NaN))?  Judged with six
    // orders of magnitude to spare on the ray directions; a hint for trace_shadow_blockers only (the closest-hit walks detect the NaN itself).
    s->view.compat &= ~RR_VIEW_NAN_BALLS;
    for (uint32_t i = 0; i < n; i++) {
        const DItem& it = s->h_items[i];
        if (!(it.flags & RR_IF_SPHERE)) continue;
        const float4 rows[3] = {it.inv0, it.inv1, it.inv2};
        double nmax = 0.0, tmax = 0.0;
        for (int r = 0; r < 3; r++) {
            nmax = std::max(nmax, std::fabs((double)rows[r].x) + std::fabs((double)rows[r].y) + std::fabs((double)rows[r].z));
            tmax = std::max(tmax, std::fabs((double)rows[r].w));
        }
        const double reach = std::max(s->tlas_reach[0], std::max(s->tlas_reach[1], s->tlas_reach[2]));
        const double on = nmax * reach + tmax, dn = nmax * 1e6, rad = std::fabs((double)it.radius);
        const bool affine = it.inv3.x == 0.0f && it.inv3.y == 0.0f && it.inv3.z == 0.0f && it.inv3.w == 1.0f;
        if (!affine || !(on * dn < 1e18) || !(rad * dn < 1e18) || !(on < 1e18) || !(rad < 1e18)) s->view.compat |= RR_VIEW_NAN_BALLS;
    }
    std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
    s->h_item_boxes.resize(4 * (size_t)n); // [0, 2n): corner boxes (lo, hi); [2n, 4n): surface boxes
    for (uint32_t i = 0; i < n; i++) {
        padded_world_box(s->h_items[i], exact[i], s->tlas_reach, &lo[3 * (size_t)i], &hi[3 * (size_t)i]);
        s->h_item_boxes[2 * (size_t)i] = make_float4(lo[3 * (size_t)i], lo[3 * (size_t)i + 1], lo[3 * (size_t)i + 2], 0.0f);
        s->h_item_boxes[2 * (size_t)i + 1] = make_float4(hi[3 * (size_t)i], hi[3 * (size_t)i + 1], hi[3 * (size_t)i + 2], 0.0f);
        float tl[3], th[3];
        padded_world_box(s->h_items[i], surf[i], s->tlas_reach, tl, th);
        s->h_item_boxes[2 * ((size_t)n + i)] = make_float4(tl[0], tl[1], tl[2], 0.0f);
        s->h_item_boxes[2 * ((size_t)n + i) + 1] = make_float4(th[0], th[1], th[2], 0.0f);
    }
    int rc = tlas_tree(s, lo.data(), hi.data(), n, &trees->corner, &trees->root);
    if (rc != RR_OK) return rc;
    // The per-ray closest-hit walks get a tree of their own over the SURFACE boxes (the argument above holds for any closest-hit query: an
    // item matters through its nearest hit alone; until round 4 only the packet form used them): fewer items are set up per ray and fewer
    // mesh walks entered.  Shadow queries keep the tree over the corner boxes (their order is the local boxes' entry distance).
    bool differs = false;
    for (size_t k = 0; k < 2 * (size_t)n && !differs; k++)
        differs = memcmp(&s->h_item_boxes[k], &s->h_item_boxes[2 * (size_t)n + k], sizeof(float4)) != 0;
    trees->has_surface = differs;
    if (differs) {
        for (uint32_t i = 0; i < n; i++) {
            const float4 tl = s->h_item_boxes[2 * ((size_t)n + i)], th = s->h_item_boxes[2 * ((size_t)n + i) + 1];
            lo[3 * (size_t)i] = tl.x; lo[3 * (size_t)i + 1] = tl.y; lo[3 * (size_t)i + 2] = tl.z;
            hi[3 * (size_t)i] = th.x; hi[3 * (size_t)i + 1] = th.y; hi[3 * (size_t)i + 2] = th.z;
        }
        rc = tlas_tree(s, lo.data(), hi.data(), n, &trees->surface, &trees->root_surface);
        if (rc != RR_OK) return rc;
    }
    return RR_OK;
}

// One top-level tree over the items' boxes lo / hi (n * 3 floats): binned SAH, one item per leaf, collapsed to 4-wide nodes.
static int tlas_tree(rr_scene* s, const float* lo, const float* hi, uint32_t n, std::vector<DNode4>* tlas4, int32_t* root4) {
    tlas4->clear();
    rr::BvhResult r;
    if (!rr::build_bvh(lo, hi, n, 1, s->tlas_depth_limit, &r))
        return fail(RR_ERR_UNSUPPORTED, "internal: top level over %u items does not fit %d levels", n, s->tlas_depth_limit);
    // leaves must name item indices directly: leaf order is a permutation, so re-code each 1-item leaf
    for (DNode& nd : r.nodes) {
        int32_t c[2];
        memcpy(&c[0], &nd.n3.x, 4); memcpy(&c[1], &nd.n3.y, 4);
        for (int k = 0; k < 2; k++)
            if (c[k] < 0) { uint32_t first = RR_LEAF_FIRST(~c[k]); c[k] = ~(int32_t)r.order[first]; }
        memcpy(&nd.n3.x, &c[0], 4); memcpy(&nd.n3.y, &c[1], 4);
    }
    if (r.root < 0) r.root = ~(int32_t)r.order[RR_LEAF_FIRST(~r.root)];
    // the form the kernels walk: collapsed to 4-wide nodes within the top level's share of the traversal stack
    int pending = 0;
    *root4 = rr::collapse_bvh4(r, s->tlas_depth_limit, false, tlas4, &pending);
    if (pending > s->tlas_depth_limit) return fail(RR_ERR_UNSUPPORTED, "top level: BVH4 stack bound exceeded");
    // Balls before meshes among the children of a node.  A walk takes the children of a node nearest box first and, at equal entry
    // distance, in slot order -- and equal is the rule where it matters: a ray that starts inside an environment sphere AND inside an
    // object's box (every secondary ray of such a scene) enters both at distance 0.  A ball is decided by a dozen instructions and
    // its toi then bounds the mesh walk that follows (closest_item passes the best hit so far down); the other way round the mesh is
    // walked without a bound first.  helmet_syn's secondary rays all end on its solid environment sphere at toi 0: with the sphere
    // in front, the walk of the 80 k-triangle mesh ends at its root.  The candidate SET and the result do not depend on the order.
    for (DNode4& nd : *tlas4) {
        int32_t code[4];
        memcpy(code, &nd.q[6], 16);
        auto is_ball = [&](int k) { return code[k] < 0 && code[k] != (int32_t)0x80000000 && (s->h_items[RR_LEAF_FIRST((uint32_t)~code[k])].flags & RR_IF_SPHERE) != 0u; };
        int order[4], m = 0;
        for (int k = 0; k < 4; k++) if (is_ball(k)) order[m++] = k;
        if (m == 0) continue;
        for (int k = 0; k < 4; k++) if (!is_ball(k) && code[k] != (int32_t)0x80000000) order[m++] = k;
        for (int k = 0; k < 4; k++) if (code[k] == (int32_t)0x80000000) order[m++] = k;
        DNode4 src = nd;
        for (int r = 0; r < 7; r++) {
            const float v[4] = {src.q[r].x, src.q[r].y, src.q[r].z, src.q[r].w};
            nd.q[r] = make_float4(v[order[0]], v[order[1]], v[order[2]], v[order[3]]);
        }
    }
    return RR_OK;
}

// The two trees into the scene's node buffer (the corner tree in its first half, the surface tree in the second: tlas_node_capacity
// nodes each), the item boxes, and the roots into the view.  Blocking copies.
static int upload_tlas(rr_scene* s, const TlasTrees& t) {
    if (t.corner.size() > s->tlas_node_capacity || t.surface.size() > s->tlas_node_capacity)
        return fail(RR_ERR_DEVICE, "top-level rebuild needs %zu / %zu nodes, capacity %u", t.corner.size(), t.surface.size(), s->tlas_node_capacity);
    if (!t.corner.empty()) HIP_TRY(hipMemcpy(s->tnodes4.p, t.corner.data(), t.corner.size() * sizeof(DNode4), hipMemcpyHostToDevice));
    if (!t.surface.empty()) HIP_TRY(hipMemcpy(s->tnodes4.as<DNode4>() + s->tlas_node_capacity, t.surface.data(), t.surface.size() * sizeof(DNode4), hipMemcpyHostToDevice));
    if (!s->h_item_boxes.empty()) HIP_TRY(hipMemcpy(s->item_boxes.p, s->h_item_boxes.data(), s->h_item_boxes.size() * sizeof(float4), hipMemcpyHostToDevice));
    s->view.tlas_root4 = t.root;
    s->view.tnodes4c = t.has_surface ? s->tnodes4.as<DNode4>() + s->tlas_node_capacity : s->tnodes4.as<DNode4>();
    s->view.tlas_root4c = t.has_surface ? t.root_surface : t.root;
    return RR_OK;
}

// Ray origins of the coming launch reach out to +-need: rebuilds the top level when its boxes were padded for less
// (a camera far outside the scene), or for more than 16x as much (the camera came back).
static int ensure_tlas_reach(rr_scene* s, const double need[3]) {
    bool grow = false, shrink = false;
    for (int c = 0; c < 3; c++) {
        if (need[c] > s->tlas_reach[c]) grow = true;
        if (s->tlas_reach[c] > 16.0 * std::max(need[c], s->tlas_floor[c])) shrink = true;
    }
    if (!grow && !shrink) return RR_OK;
    double want[3];
    for (int c = 0; c < 3; c++) want[c] = 2.0 * need[c]; // build_tlas raises it to the items' own extent
    TlasTrees trees;
    int rc = build_tlas(s, want, &trees);
    if (rc != RR_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return upload_tlas(s, trees);
}
// bound on the primary-ray origins of a camera (primary_ray: view_inv * (proj_inv * (sx, sy, -1, 1)).xyz1, |sx|, |sy| <= smax)
static void camera_reach(const rr_camera* cam, const rr_config* cfg, double need[3]) {
    const double aperture = cfg ? std::max(1.0, (double)cfg->aperture_size) : 1.0;
    const double smax = 1.0 + 2.0 * (1.0 + aperture * cam->width / 800.0) * (2.0 / std::max(1u, std::min(cam->width, cam->height)));
    const double v[4] = {smax, smax, 1.0, 1.0};
    double pp[3];
    for (int k = 0; k < 3; k++) {
        pp[k] = 0.0;
        for (int j = 0; j < 4; j++) pp[k] += std::fabs((double)cam->projection_inverse[4 * j + k]) * v[j];
    }
    for (int c = 0; c < 3; c++) {
        double m = std::fabs((double)cam->view_inverse[12 + c]);
        for (int k = 0; k < 3; k++) m += std::fabs((double)cam->view_inverse[4 * k + c]) * pp[k];
        need[c] = m * 1.001;
    }
}

// DSceneView::flat_normals from the items and triangles on the device (k_world_normals); after every upload of the items' transforms
// ... and the extent of every item's surface along its transform's rows (k_item_spans -> s->h_spans, for the top level's surface boxes).
// Everything that depends on the transforms and on the meshes is derived HERE, on the device, where the meshes are resident: the one
// blocking copy of 72 B per item at the end is the call's only wait.
static int derive_from_transforms(rr_scene* s) {
    const uint32_t n = (uint32_t)s->h_items.size();
    s->h_spans.clear();
    if (n == 0) return RR_OK;
    if (s->h_chunk_item.empty()) { // the chunk map depends on the items' triangle counts only: laid out once
        std::vector<uint2> chunks;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t nt = (s->h_items[i].flags & RR_IF_SPHERE) ? 0u : s->h_items[i].n_tris;
            for (uint32_t first = 0; first == 0u || first < nt; first += RR_ITEM_CHUNK) { chunks.push_back(make_uint2(i, first)); s->h_chunk_item.push_back(i); }
        }
        HIP_TRY(s->item_chunks.reserve(chunks.size() * sizeof(uint2)));
        HIP_TRY(hipMemcpy(s->item_chunks.p, chunks.data(), chunks.size() * sizeof(uint2), hipMemcpyHostToDevice));
        HIP_TRY(s->spans.reserve(9 * sizeof(double) * chunks.size()));
    }
    const size_t nc = s->h_chunk_item.size();
    if (nc > 0x7fffffffull) return fail(RR_ERR_UNSUPPORTED, "%zu chunks of instanced triangles", nc);
    hipLaunchKernelGGL(k_world_normals, dim3((uint32_t)nc), dim3(RR_BLOCK), 0, nullptr, s->items.as<DItem>(), s->item_chunks.as<uint2>(), s->tris.as<DTri>(), s->flat_normals.as<float4>());
    hipLaunchKernelGGL(k_item_spans, dim3((uint32_t)nc), dim3(RR_BLOCK), 0, nullptr, s->items.as<DItem>(), s->item_chunks.as<uint2>(), s->tris.as<DTri>(), s->spans.as<double>());
    HIP_TRY(hipGetLastError());
    std::vector<double> part(9 * nc);
    HIP_TRY(hipMemcpy(part.data(), s->spans.p, 9 * sizeof(double) * nc, hipMemcpyDeviceToHost)); // (waits for both kernels)
    s->h_spans.resize(9 * (size_t)n);
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 9; k++) s->h_spans[9 * (size_t)i + k] = k < 3 ? std::numeric_limits<double>::infinity() : (k < 6 ? -std::numeric_limits<double>::infinity() : 0.0);
    for (size_t c = 0; c < nc; c++) {
        double* d = &s->h_spans[9 * (size_t)s->h_chunk_item[c]];
        const double* q = &part[9 * c];
        for (int k = 0; k < 9; k++) {
            if (q[k] != q[k]) d[k] = q[k];                       // a NaN chunk poisons the item (the corner box is kept for it)
            else if (d[k] == d[k]) d[k] = k < 3 ? std::min(d[k], q[k]) : std::max(d[k], q[k]);
        }
    }
    return RR_OK;
}

// Per-triangle constants of the shading (DTri::v1.w, v3), with the IEEE binary32 sequence of rr_math.h's cross3 / dot3 / norm3 /
// normalize3 as k_shade evaluated them per hit (this file is built without contraction and without fast-math on the host side too;
// sqrtf and the division are correctly rounded on both; tests/test_gpu_math.py compares the two builds bit for bit):
//   area = norm3(cross3(a - b, a - c))            Mesh::get_normal / get_uv, src/shape/mesh.rs:127-143 (area_weights)
//   ng   = normalize3(cross3(b - a, c - a))       the triangle's own normal, src/shape/mesh.rs:76-98
static void tri_shading_constants(const float* a, const float* b, const float* c, float* ng, float* area) {
    auto cross = [](const float* u, const float* v, float* r) {
        r[0] = u[1] * v[2] - u[2] * v[1]; r[1] = u[2] * v[0] - u[0] * v[2]; r[2] = u[0] * v[1] - u[1] * v[0];
    };
    auto norm = [](const float* u) { return sqrtf((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]); };
    const float amb[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]}, amc[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
    float x[3];
    cross(amb, amc, x);
    *area = norm(x);
    const float bma[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, cma[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    cross(bma, cma, x);
    const float n = norm(x);
    ng[0] = x[0] / n; ng[1] = x[1] / n; ng[2] = x[2] / n;
}

// The binary trees of the meshes are independent: built by a few host threads (a scene of 194 meshes / 559 k triangles:
// 0.4 s on one core).  The workers pull mesh indices from one counter, so threads that could not be started only mean
// fewer hands; an exception in any worker (the builder's vectors are sized by the caller's meshes) is rethrown here.
static void build_mesh_trees(const rr_flat_scene* fs, int depth_limit, std::vector<rr::BvhResult>* built, std::vector<char>* built_ok) {
    std::atomic<uint32_t> next_mesh{0};
    auto worker = [&]() {
        for (;;) {
            const uint32_t mi = next_mesh.fetch_add(1);
            if (mi >= fs->n_meshes) break;
            RR_FAULT_POINT("scene_create.mesh_worker");
            const rr_mesh& m = fs->meshes[mi];
            const uint32_t nt = m.n_triangles;
            std::vector<float> lo(3 * (size_t)nt), hi(3 * (size_t)nt);
            for (uint32_t f = 0; f < nt; f++)
                for (int k = 0; k < 3; k++) {
                    float a = m.positions[3 * (size_t)m.indices[3 * (size_t)f] + k];
                    float b = m.positions[3 * (size_t)m.indices[3 * (size_t)f + 1] + k];
                    float c = m.positions[3 * (size_t)m.indices[3 * (size_t)f + 2] + k];
                    lo[3 * (size_t)f + k] = std::min(a, std::min(b, c));
                    hi[3 * (size_t)f + k] = std::max(a, std::max(b, c));
                }
            (*built_ok)[mi] = rr::build_bvh(lo.data(), hi.data(), nt, RR_MAX_LEAF_TRIS, depth_limit, &(*built)[mi]) ? 1 : 0;
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const uint32_t n_threads = std::min<uint32_t>(std::min<uint32_t>(hw ? hw : 4u, 16u), std::max<uint32_t>(fs->n_meshes, 1u));
    Workers pool;
    for (uint32_t t = 1; t < n_threads; t++)
        if (!pool.spawn(worker)) break;
    pool.run(worker);
    pool.join_and_rethrow();
}

// Test-only (tests/test_abi.py; not in the header): the host half of rr_scene_create -- validation and the threaded mesh tree
// builds -- without a device, so that the no-throw guard and the worker net can be exercised on a CPU-only box.
extern "C" int rr_test_host_build(const rr_flat_scene* fs, uint64_t* n_nodes_out) try {
    int rc = validate_scene(fs);
    if (rc != RR_OK) return rc;
    RR_FAULT_POINT("scene_create.host");
    std::vector<rr::BvhResult> built(fs->n_meshes);
    std::vector<char> built_ok(fs->n_meshes, 0);
    build_mesh_trees(fs, RR_BLAS_MAX_DEPTH, &built, &built_ok);
    uint64_t n = 0;
    for (uint32_t mi = 0; mi < fs->n_meshes; mi++) {
        if (!built_ok[mi]) return fail(RR_ERR_UNSUPPORTED, "mesh %u: BVH depth limit exceeded", mi);
        n += built[mi].nodes.size();
    }
    if (n_nodes_out) *n_nodes_out = n;
    return RR_OK;
} RR_GUARD_END("rr_test_host_build")

extern "C" int rr_scene_create(const rr_flat_scene* fs, int device, rr_scene** out) try {
    if (!out) return fail(RR_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int rc = validate_scene(fs);
    if (rc != RR_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RR_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(RR_ERR_INVALID_ARGUMENT, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<rr_scene> s(new rr_scene);
    RR_FAULT_POINT("scene_create.host");
    s->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // u8 -> f32 table, exactly (float)i / 255.0f
    float lut[256];
    for (int i = 0; i < 256; i++) lut[i] = (float)i / 255.0f;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_u8_to_f32), lut, sizeof lut));

    // ---- textures: one RGBA8 pool
    std::vector<DTexture> dtex(fs->n_textures);
    uint64_t n_texels = 0;
    for (uint32_t i = 0; i < fs->n_textures; i++) {
        dtex[i].offset = n_texels; dtex[i].width = fs->textures[i].width; dtex[i].height = fs->textures[i].height;
        n_texels += (uint64_t)fs->textures[i].width * fs->textures[i].height;
    }
    HIP_TRY(s->texels.reserve(std::max<uint64_t>(n_texels, 1) * 4));
    for (uint32_t i = 0; i < fs->n_textures; i++) {
        uint64_t n = (uint64_t)dtex[i].width * dtex[i].height;
        if (n) HIP_TRY(hipMemcpy(s->texels.as<uint32_t>() + dtex[i].offset, fs->textures[i].rgba8, n * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(s->textures.reserve(std::max<size_t>(dtex.size(), 1) * sizeof(DTexture)));
    if (!dtex.empty()) HIP_TRY(hipMemcpy(s->textures.p, dtex.data(), dtex.size() * sizeof(DTexture), hipMemcpyHostToDevice));

    // ---- materials
    s->tex_width.resize(fs->n_textures);
    for (uint32_t i = 0; i < fs->n_textures; i++) s->tex_width[i] = fs->textures[i].width;
    std::vector<DMaterial> dmat(fs->n_materials);
    s->h_textures = dtex;
    for (uint32_t i = 0; i < fs->n_materials; i++) dmat[i] = make_dmaterial(fs->materials[i], s->tex_width, s->h_textures);
    HIP_TRY(s->materials.reserve(std::max<size_t>(dmat.size(), 1) * sizeof(DMaterial)));
    if (!dmat.empty()) HIP_TRY(hipMemcpy(s->materials.p, dmat.data(), dmat.size() * sizeof(DMaterial), hipMemcpyHostToDevice));

    // ---- lights (disabled lights keep their slot: the slot is the RNG stream of their shadow jitter)
    std::vector<DLight> dl(fs->n_lights);
    s->n_enabled_lights = 0;
    for (uint32_t i = 0; i < fs->n_lights; i++) {
        const rr_light& l = fs->lights[i];
        for (int k = 0; k < 3; k++) { dl[i].pos[k] = l.pos[k]; dl[i].dir[k] = l.dir[k]; dl[i].color[k] = l.color[k]; }
        dl[i].intensity = l.intensity; dl[i].max_angle = l.max_angle;
        dl[i].type = l.light_type | (l.enabled ? 0u : 0x80u);
        if (l.enabled) s->n_enabled_lights++;
    }
    HIP_TRY(s->lights.reserve(std::max<size_t>(dl.size(), 1) * sizeof(DLight)));
    if (!dl.empty()) HIP_TRY(hipMemcpy(s->lights.p, dl.data(), dl.size() * sizeof(DLight), hipMemcpyHostToDevice));

    // ---- shares of the traversal stack (RR_STACK_DEPTH entries per lane): a top level over n items never needs more
    // than n - 1 pending entries, so a scene of few items leaves more levels to its per-mesh trees (a 320 k-triangle
    // mesh traces 3 % faster with 30 levels than with 24, and 7 % slower with 20)
    s->tlas_depth_limit = (int)std::min<uint32_t>(RR_TLAS_MAX_DEPTH, std::max<uint32_t>(1u, fs->n_items > 1 ? fs->n_items - 1 : 1u));
    // More than 2^RR_TLAS_MAX_DEPTH items (the reference has no limit: `items: Vec<..>`, src/scene.rs:69-83): the top level takes the
    // levels it needs -- ceil(log2 n): the builder falls back to object-median splits where the budget gets tight -- out of the
    // per-mesh trees' share, down to 16 levels for those (8 * 2^16 triangles per mesh at worst); RR_MAX_ITEMS = 2^20 is where that ends.
    if (fs->n_items > (1u << RR_TLAS_MAX_DEPTH)) {
        if (fs->n_items > RR_MAX_ITEMS) return fail(RR_ERR_UNSUPPORTED, "%u items (RR_MAX_ITEMS = %u)", fs->n_items, RR_MAX_ITEMS);
        int need = RR_TLAS_MAX_DEPTH;
        while ((1u << need) < fs->n_items) need++;
        s->tlas_depth_limit = need;
    }
    s->blas_depth_limit = RR_STACK_DEPTH - 3 - s->tlas_depth_limit;
    // ---- meshes: one BLAS per mesh, shared by every item that names it
    struct MeshDev { uint32_t tri_base, n_tris; uint32_t node_base4; int32_t root4; bool has_normals, degenerate; };
    std::vector<MeshDev> md(fs->n_meshes);
    std::vector<DNode4> all_nodes4;
    std::vector<DTri> all_tris;
    std::vector<DTriX> all_trix;
    std::vector<DTriAttr> all_attrs;
    std::vector<uint32_t> all_face_slot;
    // the binary trees of the meshes are independent: built by a few host threads (a scene of 194 meshes / 559 k triangles:
    // 0.4 s on one core), then collapsed and laid out one after the other
    std::vector<rr::BvhResult> built(fs->n_meshes);
    std::vector<char> built_ok(fs->n_meshes, 0);
    build_mesh_trees(fs, s->blas_depth_limit, &built, &built_ok);
    for (uint32_t mi = 0; mi < fs->n_meshes; mi++) {
        const rr_mesh& m = fs->meshes[mi];
        uint32_t nt = m.n_triangles;
        if (!built_ok[mi]) return fail(RR_ERR_UNSUPPORTED, "mesh %u: %u triangles need a deeper tree than the %d levels left beside a top level over %u items",
                                       mi, nt, s->blas_depth_limit, fs->n_items);
        rr::BvhResult& r = built[mi];
        md[mi].tri_base = (uint32_t)all_tris.size();
        md[mi].n_tris = nt;
        md[mi].has_normals = m.n_normals > 0 && m.n_normal_faces > 0;
        md[mi].degenerate = false;
        {
            int pending = 0;
            md[mi].node_base4 = (uint32_t)all_nodes4.size();
            md[mi].root4 = rr::collapse_bvh4(r, s->blas_depth_limit, true, &all_nodes4, &pending);
            if (pending > s->blas_depth_limit) return fail(RR_ERR_UNSUPPORTED, "mesh %u: BVH4 stack bound exceeded", mi);
        }
        size_t fs_base = all_face_slot.size();
        all_face_slot.resize(fs_base + nt);
        for (uint32_t slot = 0; slot < nt; slot++) {
            uint32_t f = r.order[slot];
            all_face_slot[fs_base + f] = slot;
            s->h_slot_face.push_back(f);
            const uint32_t* ix = m.indices + 3 * (size_t)f;
            const float *a = m.positions + 3 * (size_t)ix[0], *b = m.positions + 3 * (size_t)ix[1], *c = m.positions + 3 * (size_t)ix[2];
            {   // Mesh::get_uv divides by the triangle's area (src/shape/mesh.rs:127-143): a zero (or non-finite) area makes the uv of ANY point
                // on that face non-finite.  Judged in double with a generous margin: a false positive only costs shadow rays that a
                // zero light term would have skipped (k_shade, want_shadow)
                const double u[3] = {(double)a[0] - b[0], (double)a[1] - b[1], (double)a[2] - b[2]}, v[3] = {(double)a[0] - c[0], (double)a[1] - c[1], (double)a[2] - c[2]};
                const double cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
                const double area2 = cx * cx + cy * cy + cz * cz, scale2 = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (!(area2 > 1e-10 * scale2) || !(area2 > 1e-24) || !std::isfinite(area2)) md[mi].degenerate = true;
            }
            DTri t;
            float fbits; memcpy(&fbits, &f, 4);
            float ng[3], area;
            tri_shading_constants(a, b, c, ng, &area);
            t.v0 = make_float4(a[0], a[1], a[2], fbits);
            t.v1 = make_float4(b[0], b[1], b[2], area);
            t.v2 = make_float4(c[0], c[1], c[2], 0.0f);
            t.v3 = make_float4(ng[0], ng[1], ng[2], 0.0f);
            all_tris.push_back(t);
            {   // the edge vectors parry's test evaluates per ray: ab = b - a, ac = c - a
                const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                DTriX x;
                x.t0 = t.v0;
                x.t1 = make_float4(ab[0], ab[1], ab[2], ac[0]);
                x.t2 = make_float4(ac[1], ac[2], 0.0f, 0.0f);
                all_trix.push_back(x);
            }
            DTriAttr at;
            float n[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
            if (md[mi].has_normals)
                for (int v = 0; v < 3; v++)
                    for (int k = 0; k < 3; k++) n[v][k] = m.normals[3 * (size_t)m.normal_indices[3 * (size_t)f + v] + k];
            uint32_t flags = 0;
            if (f < m.n_uv_faces) { // Mesh::get_uv bounds test, reference src/shape/mesh.rs:116
                flags |= 1u;
                for (int v = 0; v < 3; v++)
                    for (int k = 0; k < 2; k++) uv[v][k] = m.uvs[2 * (size_t)m.uv_indices[3 * (size_t)f + v] + k];
            }
            float flb; memcpy(&flb, &flags, 4);
            at.s0 = make_float4(n[0][0], n[0][1], n[0][2], uv[0][0]);
            at.s1 = make_float4(n[1][0], n[1][1], n[1][2], uv[0][1]);
            at.s2 = make_float4(n[2][0], n[2][1], n[2][2], uv[1][0]);
            at.s3 = make_float4(uv[1][1], uv[2][0], uv[2][1], flb);
            all_attrs.push_back(at);
        }
    }

    // ---- items
    s->h_items.resize(fs->n_items);
    s->item_host.resize(fs->n_items);
    s->n_materials = fs->n_materials;
    bool general_w = false;
    uint64_t n_flat_normals = 0; // entries of DSceneView::flat_normals: two per instanced triangle
    for (uint32_t i = 0; i < fs->n_items; i++) {
        const rr_item& it = fs->items[i];
        const rr_material& cache = fs->materials[it.material_cache];
        const rr_material& full = fs->materials[it.material];
        DItem& d = s->h_items[i];
        memset(&d, 0, sizeof d);
        fill_item_matrices(d, it.trans, it.trans_inv);
        if (!(it.trans_inv[3] == 0.0f && it.trans_inv[7] == 0.0f && it.trans_inv[11] == 0.0f && it.trans_inv[15] == 1.0f)) general_w = true;
        for (int k = 0; k < 3; k++) { d.bmin[k] = it.bbox_min[k]; d.bmax[k] = it.bbox_max[k]; }
        d.radius = it.radius;
        d.id = it.id;
        d.material = it.material;
        ItemHost& ih = s->item_host[i];
        ih = ItemHost{it.kind, it.material, it.material_cache, it.visible != 0, it.flip_normals != 0, false, false, it.kind != RR_ITEM_SPHERE ? (int32_t)it.mesh : -1};
        if (it.kind != RR_ITEM_SPHERE) {
            const MeshDev& m = md[it.mesh];
            d.tri_base = m.tri_base; d.n_tris = m.n_tris;
            d.node_base4 = m.node_base4; d.root4 = m.root4;
            if (n_flat_normals + 2ull * m.n_tris > 0xffffffffull) return fail(RR_ERR_UNSUPPORTED, "more than 2^31 instanced triangles");
            d.wn_base = (uint32_t)n_flat_normals; n_flat_normals += 2ull * m.n_tris;
            ih.mesh_has_normals = m.has_normals; ih.mesh_degenerate = m.degenerate;
        }
        const uint32_t f = item_flags(ih, cache, full, s->tex_width);
        d.flags = f;
        if (f & RR_IF_OCCLUDER_ALPHA_TEX) s->view.any_alpha_occluder = 1u;
    }

    auto upload = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.reserve(std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    if (all_nodes4.size() >= (1u << 25)) return fail(RR_ERR_UNSUPPORTED, "%zu BVH4 nodes (nodes are addressed with 32-bit byte offsets)", all_nodes4.size());
    HIP_TRY(upload(s->nodes4, all_nodes4.data(), all_nodes4.size() * sizeof(DNode4)));
    HIP_TRY(upload(s->tris, all_tris.data(), all_tris.size() * sizeof(DTri)));
    if (all_trix.size() >= (1u << 26)) return fail(RR_ERR_UNSUPPORTED, "%zu triangles (addressed with 32-bit byte offsets)", all_trix.size());
    static_assert(sizeof(DTriX) == 48 && sizeof(DNode4) == 128 && sizeof(DMaterial) == 240, "layouts the kernels address by byte offset");
    HIP_TRY(upload(s->trix, all_trix.data(), all_trix.size() * sizeof(DTriX)));
    HIP_TRY(upload(s->attrs, all_attrs.data(), all_attrs.size() * sizeof(DTriAttr)));
    HIP_TRY(upload(s->face_slot, all_face_slot.data(), all_face_slot.size() * 4));
    HIP_TRY(upload(s->items, s->h_items.data(), s->h_items.size() * sizeof(DItem)));
    HIP_TRY(s->flat_normals.reserve(std::max<size_t>((size_t)n_flat_normals * sizeof(float4), 16)));
    rc = derive_from_transforms(s.get()); // flat world normals; the extent of every item's surface, for the top level below
    if (rc != RR_OK) return rc;

    // ---- top level: always present (even for one item), so the kernels have a single traversal path.
    // The reference's choice between "all items" and its scene BVH (src/raytracing.rs:434) only changes the
    // candidate set, never the result.
    {
        TlasTrees trees;
        const double none[3] = {0.0, 0.0, 0.0};
        rc = build_tlas(s.get(), none, &trees);
        if (rc != RR_OK) return rc;
        for (int c = 0; c < 3; c++) s->tlas_floor[c] = s->tlas_reach[c];
        s->tlas_node_capacity = std::max<uint32_t>((uint32_t)std::max(trees.corner.size(), trees.surface.size()), fs->n_items ? fs->n_items : 1u); // room for rebuilds after transform updates
        HIP_TRY(s->tnodes4.reserve(2 * (size_t)s->tlas_node_capacity * sizeof(DNode4)));
        HIP_TRY(hipMemset(s->tnodes4.p, 0, 2 * (size_t)s->tlas_node_capacity * sizeof(DNode4)));
        HIP_TRY(s->item_boxes.reserve(std::max<size_t>(s->h_item_boxes.size() * sizeof(float4), 16)));
        s->view.tnodes4 = s->tnodes4.as<DNode4>();
        rc = upload_tlas(s.get(), trees);
        if (rc != RR_OK) return rc;
    }

    DSceneView& v = s->view;
    v.flat_normals = s->flat_normals.as<float4>();
    v.items = s->items.as<DItem>(); v.nodes4 = s->nodes4.as<DNode4>(); v.tris = s->tris.as<DTri>(); v.trix = s->trix.as<DTriX>(); v.attrs = s->attrs.as<DTriAttr>();
    v.face_slot = s->face_slot.as<uint32_t>();
    v.materials = s->materials.as<DMaterial>(); v.textures = s->textures.as<DTexture>(); v.texels = s->texels.as<uint32_t>();
    v.lights = s->lights.as<DLight>();
    v.n_items = fs->n_items; v.n_lights = fs->n_lights; v.n_enabled_lights = s->n_enabled_lights;
    v.item_boxes = s->item_boxes.as<float4>();
    v.general_w = general_w ? 1u : 0u;

    HIP_TRY(s->pool.reserve(POOL_WORDS * 4));
    HIP_TRY(s->counters.reserve(RR_CNT_WORDS * 8));
    HIP_TRY(hipEventCreate(&s->frame_a));
    HIP_TRY(hipEventCreate(&s->frame_b));
    HIP_TRY(hipEventCreateWithFlags(&s->count_ready, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&s->h_count, 64, hipHostMallocDefault));
    *out = s.release();
    return RR_OK;
} RR_GUARD_END("rr_scene_create")

extern "C" void rr_scene_destroy(rr_scene* s) {
    if (!s) return;
    try {
        (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();
        delete s; // ~rr_scene: events, pinned memory; ~DevBuf: every device buffer
    } catch (...) { (void)guard_fail("rr_scene_destroy"); }
}

extern "C" int rr_scene_update_transforms(rr_scene* s, const float* trans, const float* trans_inv) try {
    if (!s || !trans || !trans_inv) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    RR_FAULT_POINT("update_transforms.host");
    uint32_t n = (uint32_t)s->h_items.size();
    bool general_w = false;
    for (uint32_t i = 0; i < n; i++) {
        const float *t = trans + 16 * (size_t)i, *ti = trans_inv + 16 * (size_t)i;
        if (!finite16(t) || !finite16(ti)) return fail(RR_ERR_INVALID_ARGUMENT, "item %u: non-finite transform", i);
        fill_item_matrices(s->h_items[i], t, ti);
        if (!(ti[3] == 0.0f && ti[7] == 0.0f && ti[11] == 0.0f && ti[15] == 1.0f)) general_w = true;
    }
    HIP_TRY(hipDeviceSynchronize()); // no frame may be in flight on the records that change (a caller that renders asynchronously through rr_render_region_device)
    HIP_TRY(hipMemcpyAsync(s->items.p, s->h_items.data(), n * sizeof(DItem), hipMemcpyHostToDevice, nullptr));
    { int rc = derive_from_transforms(s); if (rc != RR_OK) return rc; }
    s->view.general_w = general_w ? 1u : 0u;
    {
        TlasTrees trees;
        const double none[3] = {0.0, 0.0, 0.0};
        int rc = build_tlas(s, none, &trees); // the next frame's camera grows the reach again if it has to
        if (rc != RR_OK) return rc;
        for (int c = 0; c < 3; c++) s->tlas_floor[c] = s->tlas_reach[c];
        rc = upload_tlas(s, trees);
        if (rc != RR_OK) return rc;
    }
    return RR_OK;
} RR_GUARD_END("rr_scene_update_transforms")

// Material edits between frames (GUI sliders: reference src/run.rs:1132-1133 writes through Material::apply_diff,
// src/shape/mod.rs:182-242): every material record is replaced and the item flag words derived from the material
// caches are rebuilt; geometry, acceleration structures and texture images stay as uploaded.
extern "C" int rr_scene_update_materials(rr_scene* s, const rr_material* materials, uint32_t n_materials) try {
    if (!s || !materials) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (n_materials != s->n_materials) return fail(RR_ERR_INVALID_ARGUMENT, "%u materials, the scene was created with %u", n_materials, s->n_materials);
    for (uint32_t i = 0; i < n_materials; i++)
        for (int k = 0; k < RR_TEX_COUNT; k++)
            if (materials[i].texture[k] >= (int32_t)s->tex_width.size()) return fail(RR_ERR_INVALID_ARGUMENT, "material %u texture slot %d = %d out of range", i, k, materials[i].texture[k]);
    for (const ItemHost& ih : s->item_host)
        for (int k = 0; k < RR_TEX_COUNT; k++)
            if (materials[ih.material_cache].texture[k] >= 0)
                return fail(RR_ERR_INVALID_ARGUMENT, "material %d is a material cache and must not carry textures (reference src/shape/mod.rs:769-772)", ih.material_cache);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<DMaterial> dmat(n_materials);
    for (uint32_t i = 0; i < n_materials; i++) dmat[i] = make_dmaterial(materials[i], s->tex_width, s->h_textures);
    s->view.any_alpha_occluder = 0u;
    for (size_t i = 0; i < s->item_host.size(); i++) {
        s->h_items[i].flags = item_flags(s->item_host[i], materials[s->item_host[i].material_cache], materials[s->item_host[i].material], s->tex_width);
        if (s->h_items[i].flags & RR_IF_OCCLUDER_ALPHA_TEX) s->view.any_alpha_occluder = 1u;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (n_materials) HIP_TRY(hipMemcpy(s->materials.p, dmat.data(), dmat.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
    if (!s->h_items.empty()) HIP_TRY(hipMemcpy(s->items.p, s->h_items.data(), s->h_items.size() * sizeof(DItem), hipMemcpyHostToDevice));
    return RR_OK;
} RR_GUARD_END("rr_scene_update_materials")

// ---------------------------------------------------------------------------
// frame
// ---------------------------------------------------------------------------
static int check_frame_args(const rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy) {
    if (!s || !cam || !cfg) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (cfg->samples == 0) return fail(RR_ERR_INVALID_ARGUMENT, "samples must be >= 1");
    // with the caller's table the reference's own u16 limit applies; the built-in table stops where its shuffle stays affordable
    if (cfg->samples > (sample_xy ? RR_MAX_SAMPLES_WITH_TABLE : RR_MAX_SAMPLES))
        return fail(RR_ERR_UNSUPPORTED, "samples %u > %u%s", (unsigned)cfg->samples, sample_xy ? RR_MAX_SAMPLES_WITH_TABLE : RR_MAX_SAMPLES,
                    sample_xy ? "" : " (the built-in sub-sample table; pass sample_xy for up to 32766)");
    if (cfg->max_recursion > RR_MAX_RECURSION) return fail(RR_ERR_UNSUPPORTED, "max_recursion %u > %u", cfg->max_recursion, RR_MAX_RECURSION);
    if (cam->width == 0 || cam->height == 0 || cam->width > 65535u || cam->height > 65535u) return fail(RR_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", cam->width, cam->height);
    if (!finite16(cam->projection_inverse) || !finite16(cam->view_inverse)) return fail(RR_ERR_INVALID_ARGUMENT, "non-finite camera matrix");
    return RR_OK;
}

static hipEvent_t take_event(rr_scene* s) {
    if (!s->event_pool.empty()) { hipEvent_t e = s->event_pool.back(); s->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ScopedTimer {
    rr_scene* s; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(rr_scene* s_, hipStream_t st_, int kind_) : s(s_), st(st_), kind(kind_) {
        if (s->profiling) { a = take_event(s); b = take_event(s); (void)hipEventRecord(a, st); }
    }
    ~ScopedTimer() { if (s->profiling) { (void)hipEventRecord(b, st); s->timed.push_back(TimedLaunch{a, b, kind}); } }
};

static void resolve_timers(rr_scene* s) {
    for (auto& t : s->timed) {
        float ms = 0.0f;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            if (t.kind == 0 || t.kind == 4) { s->stats.ms_trace_closest += ms; s->stats.launches_trace_closest++; }
            if (t.kind == 4) { s->stats.ms_trace_closest_level1 += ms; s->stats.launches_trace_closest_level1++; }
            if (t.kind == 1 || t.kind == 6) { s->stats.ms_trace_shadow += ms; s->stats.launches_trace_shadow++; }
            if (t.kind == 6) { s->stats.ms_trace_shadow_level1 += ms; s->stats.launches_trace_shadow_level1++; }
            if (t.kind == 2 || t.kind == 5) { s->stats.ms_shade += ms; s->stats.launches_shade++; }
            if (t.kind == 5) { s->stats.ms_shade_level1 += ms; s->stats.launches_shade_level1++; }
            else if (t.kind == 3) { s->stats.ms_binning += ms; }
        }
        s->event_pool.push_back(t.a); s->event_pool.push_back(t.b);
    }
    s->timed.clear();
}

// Progressive preview (rr_render_progressive): after every device batch that ends on a whole slice of samples the
// accumulators are resolved over the samples finished so far and handed to the caller.
struct PassHook {
    rr_pass_fn fn; void* user; uint32_t min_passes;
    void* host[4]; size_t bytes[4];
};

// The ONE place that launches the closest-hit kernel: the frame path (run_level), rr_pick and rr_trace_rays all come through here, so a
// change to the kernel's arguments cannot leave one caller behind.  (Round 3, scratch run r3c50: a variant whose LEVEL-1 build stored
// the primary rays through q.r0 / q.r1 aborted the process inside rr_pick -- rr_pick built its own argument list with those pointers
// NULL, which is right for the kernel at HEAD, which never touches them, and was a write to address 16 * i for that variant.)
// Every pointer the build in question may touch is checked here, on the host, before the launch; level 1 reads no ray records (the
// rays are derived from their index), so its queue carries the hit records only.
static int launch_trace_closest(rr_scene* s, bool primary, DRayQueue q, uint32_t* count, uint32_t* head, uint64_t n, const DShadeConst* kc,
                                const uint32_t* slot_xy, const DPrimary& pr, unsigned long long* counters, hipStream_t st) {
    if (!count || !head || !q.hit || !kc || !counters) return fail(RR_ERR_DEVICE, "internal: closest-hit launch with a NULL argument");
    if (primary && (!slot_xy || !pr.sample_xy || pr.n != n)) return fail(RR_ERR_DEVICE, "internal: level-1 closest-hit launch without its ray table");
    if (!primary && (!q.r0 || !q.r1 || !q.r2)) return fail(RR_ERR_DEVICE, "internal: closest-hit launch without ray records");
    if (n == 0 || n > 0x7fffff00ull) return fail(RR_ERR_DEVICE, "internal: closest-hit launch of %llu rays", (unsigned long long)n);
    const int grid = (int)std::min<uint64_t>((n + RR_BLOCK - 1) / RR_BLOCK, (uint64_t)s->n_cus * RR_CLOSEST_WAVES);
    if (primary) {
        q.r0 = nullptr; q.r1 = nullptr; q.r2 = nullptr;
        hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(RR_BLOCK), 0, st, s->view, q, count, head, kc, slot_xy, pr, counters);
    } else {
        hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(RR_BLOCK), 0, st, s->view, q, count, head, kc, slot_xy, pr, counters);
    }
    HIP_TRY(hipGetLastError());
    return RR_OK;
}

static int render_region_locked(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy,
                                const rr_region* rg, const rr_frame* out, bool frame_layout, hipStream_t st, const volatile int* cancel,
                                const PassHook* hook = nullptr) {
    HIP_TRY(hipSetDevice(s->device));
    if (st != s->last_stream) { HIP_TRY(hipStreamSynchronize(s->last_stream)); s->last_stream = st; }
    const uint32_t W = cam->width, H = cam->height;
    if ((uint64_t)W * H > (1ull << 30)) return fail(RR_ERR_UNSUPPORTED, "frame of %ux%u pixels", W, H);
    // ---- region map
    if (memcmp(&s->region_cached, rg, sizeof *rg) != 0 || s->region_w != W || s->region_h != H) {
        std::vector<uint32_t> order;
        fill_region(W, H, *rg, &s->h_region_xy, &order);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(s->region_xy.reserve(std::max<size_t>(s->h_region_xy.size(), 1) * 4));
        HIP_TRY(s->trace_order.reserve(std::max<size_t>(order.size(), 1) * 4));
        if (!s->h_region_xy.empty()) {
            // slot_xy[j] = pixel of accumulator slot j; slot_out[j] = its index in the compact output order
            std::vector<uint32_t> slot_xy(order.size());
            for (size_t j = 0; j < order.size(); j++) slot_xy[j] = s->h_region_xy[order[j]];
            HIP_TRY(hipMemcpy(s->region_xy.p, slot_xy.data(), slot_xy.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(s->trace_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        }
        s->region_cached = *rg; s->region_w = W; s->region_h = H;
    }
    const uint32_t npix = (uint32_t)s->h_region_xy.size();
    resolve_timers(s); // launches of an earlier frame nobody asked about must not leak into this frame's stats
    memset(&s->stats, 0, sizeof s->stats);
    s->stats_final = false;
    if (npix == 0) return RR_OK;

    { // the top level's boxes must be padded for this camera's distance from the origin
        double need[3];
        camera_reach(cam, cfg, need);
        int rc = ensure_tlas_reach(s, need);
        if (rc != RR_OK) return rc;
    }

    // ---- frame constants
    DFrame fr;
    memset(&fr, 0, sizeof fr);
    memcpy(fr.proj_inv, cam->projection_inverse, 64);
    memcpy(fr.view_inv, cam->view_inverse, 64);
    fr.width = W; fr.height = H; fr.samples = cfg->samples; fr.cell_size = cell_size_of(cfg->samples);
    fr.max_recursion = cfg->max_recursion; fr.monte_carlo = cfg->monte_carlo ? 1u : 0u; fr.gamma = cfg->gamma_correction ? 1u : 0u;
    fr.dof = (cfg->aperture_size > 1.0f && cfg->focal_length > 1.0f) ? 1u : 0u;
    fr.focal_length = cfg->focal_length; fr.aperture_size = cfg->aperture_size; fr.fog_density = cfg->fog_density;
    for (int k = 0; k < 3; k++) fr.fog_color[k] = cfg->fog_color[k];
    fr.seed_lo = (uint32_t)cfg->seed; fr.seed_hi = (uint32_t)(cfg->seed >> 32);
    fr.n_region_pixels = npix;

    // ---- the shade kernel's constants (scene view + frame), read from device memory
    {
        DShadeConst hc;
        hc.sc = s->view; hc.fr = fr;
        HIP_TRY(s->shade_const.reserve(sizeof hc));
        HIP_TRY(hipMemcpyAsync(s->shade_const.p, &hc, sizeof hc, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st)); // `hc` is a stack local
    }

    // ---- sample table
    if (!sample_xy) { // the built-in table depends on the sample count only: built once per count, not once per frame
        if (s->table_samples != cfg->samples) {
            s->table_samples = 0; // the cache names a sample count only once its table is complete
            try { s->table_cache.resize((size_t)cfg->samples * 2); }
            catch (const std::exception&) { return fail(RR_ERR_OUT_OF_MEMORY, "no host memory for the sub-sample table"); }
            const int rct = rr_sample_table(cfg->samples, s->table_cache.data(), nullptr);
            if (rct != RR_OK) return rct;
            s->table_samples = cfg->samples;
        }
        sample_xy = s->table_cache.data();
    }
    HIP_TRY(s->sample_xy.reserve((size_t)cfg->samples * 4));
    HIP_TRY(hipMemcpyAsync(s->sample_xy.p, sample_xy, (size_t)cfg->samples * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st)); // the caller's table may be a temporary

    // ---- accumulators
    HIP_TRY(s->acc_rgb.reserve((size_t)npix * 24));
    HIP_TRY(s->acc_normal.reserve((size_t)npix * 24));
    HIP_TRY(s->acc_depth.reserve((size_t)npix * 8));
    HIP_TRY(s->acc_id.reserve((size_t)npix * 4));
    HIP_TRY(s->acc_flags.reserve((size_t)npix * 4));
    HIP_TRY(hipMemsetAsync(s->acc_flags.p, 0, (size_t)npix * 4, st));
    HIP_TRY(hipMemsetAsync(s->acc_rgb.p, 0, (size_t)npix * 24, st));
    HIP_TRY(hipMemsetAsync(s->acc_normal.p, 0, (size_t)npix * 24, st));
    HIP_TRY(hipMemsetAsync(s->acc_depth.p, 0, (size_t)npix * 8, st));
    HIP_TRY(hipMemsetAsync(s->acc_id.p, 0, (size_t)npix * 4, st));
    HIP_TRY(hipMemsetAsync(s->counters.p, 0, RR_CNT_WORDS * 8, st));
    DAccum acc{s->acc_rgb.as<long long>(), s->acc_normal.as<long long>(), s->acc_depth.as<long long>(), s->acc_id.as<uint32_t>(), (unsigned long long)npix,
               s->acc_flags.as<uint32_t>()};
    // aux outputs the caller did not ask for are not accumulated at all
    if (!out->normal) acc.normal = nullptr;
    if (!out->depth) acc.depth = nullptr;
    if (!out->object_id) acc.object_id = nullptr;

    // ---- ray memory.  All live depth levels of a batch sit in ONE arena of ray records (56 B each), level d + 1
    // stacked behind level d.  A level of n rays spawns at most 2 n children; if they fit behind it the level is
    // shaded in one go, otherwise in slices whose children fit, each slice's subtree finished (depth first) before
    // the next slice starts.  So capacity never limits correctness, only how large the launches can be -- and launch
    // size matters: the persistent trace kernels lose 8-15 % to ramp-up and tail per launch at 12 M rays
    // (reserving the worst case 2^(d-1) growth per level, as the first version did, capped batches there).
    const uint32_t R = cfg->max_recursion;
    // Arena memory: a quarter of what is free on the device, at most 64 GB (MI355X has 288 GB of HBM3E),
    // unless rr_tuning::queue_budget_bytes says otherwise.  Memory already held by this scene's arena counts as free.
    uint64_t budget;
    if (s->tuning.queue_budget_bytes) budget = s->tuning.queue_budget_bytes;
    else {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        budget = std::min<uint64_t>((free_b + 56ull * s->arena_cap + s->hit1.bytes) / 4, 64ull << 30);
    }
    const uint64_t total_primary = (uint64_t)npix * cfg->samples;
    const uint64_t LEVEL_MAX = 0x7fffff00ull; // ray indices are 32-bit
    // Level 1 (the primary rays) needs only its 16-B hit records: the rays themselves are derived from their index
    // (primary_ray).  The arena holds the deeper levels; 2 arena rays per primary ray cover every level of a typical
    // frame at once (sponza_syn: all deeper levels together hold 4 % of the primaries), sliced when a scene branches more.
    const uint64_t per_primary = 16ull + 2ull * 56ull;
    uint64_t B = std::max<uint64_t>(budget / per_primary, 4096);
    B = std::min<uint64_t>(B, std::min<uint64_t>(total_primary, LEVEL_MAX));
    if (hook && hook->min_passes > 1) B = std::min<uint64_t>(B, std::max<uint64_t>(npix, (total_primary + hook->min_passes - 1) / hook->min_passes));
    // equal batches (a frame that needs 1.2 batches would otherwise end with a small, poorly filled one)
    { const uint64_t nb = (total_primary + B - 1) / B; B = (total_primary + nb - 1) / nb; }
    if (B > npix) B = ((B + npix - 1) / npix) * npix; // whole sample slices when possible
    // Sample grouping: a packet of 64 primary rays = 64/G neighbouring pixels x G samples of each (primary_ray), so the
    // rays of a wave - and the shadow rays and children they spawn - start almost identical and their walks stay
    // together.  The largest group the sample count allows is best (closest-hit -30 % on sponza_syn at G = 64 against
    // one sample of 64 pixels), given that the wave merges its accumulator adds per pixel first (accum_merged): 64
    // lanes adding to one address otherwise cost more than the walks gain.  Needs whole groups per batch.
    uint32_t G = 1;
    {
        const uint32_t forced = s->tuning.sample_group; // 0 = automatic
        for (uint32_t g = forced ? forced : 64u; g >= 2; g >>= 1)
            if (g <= 64 && !(g & (g - 1)) && cfg->samples % g == 0 && npix % (RR_WAVE / g) == 0 && (uint64_t)npix * g <= B) { G = g; break; }
        if (forced && G != forced) G = 1;
    }
    if (G > 1) {
        // whole groups per batch, batches as equal as whole groups allow
        const uint64_t unit = (uint64_t)npix * G, units_max = B / unit, total_units = total_primary / unit;
        const uint64_t nb = (total_units + units_max - 1) / units_max;
        B = ((total_units + nb - 1) / nb) * unit;
    }
    B = std::min<uint64_t>(B, total_primary);
    // arena (levels 2 and deeper): 2 rays per primary ray, or 7 where that stays under 16 GB (a branching scene then fits
    // on its first frame too), or `arena_factor` after a frame that had to slice -- always within the budget
    const uint64_t after_hits = budget > 16ull * B ? (budget - 16ull * B) / 56ull : 0ull;
    const uint64_t roomy = std::min<uint64_t>(7 * B, (16ull << 30) / 56ull);
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(2 * B, roomy), (uint64_t)s->arena_factor * B);
    const uint64_t M = std::min<uint64_t>(std::min<uint64_t>(want, std::max<uint64_t>(2 * B, after_hits)) + 2ull * RR_BLOCK * (R + 1), LEVEL_MAX);
    const size_t elem[4] = {16, 16, 8, 16};
    if (M > s->arena_cap) {
        for (int k = 0; k < 4; k++) HIP_TRY(s->arena[k].reserve(M * elem[k]));
        s->arena_cap = M;
    }
    HIP_TRY(s->hit1.reserve(B * 16));
    // (the shadow queue holds one 48-B ray per hit of the chunk and ENABLED light: with many lights the chunk shrinks so that
    // the queue stays within 16 GB -- the reference has no limit on lights, src/raytracing.rs:814)
    const uint64_t chunk_by_lights = std::max<uint64_t>(65536, ((16ull << 30) / (48ull * std::max<uint32_t>(s->n_enabled_lights, 1u))) / (RR_BLOCK * RR_SQ_SHARDS) * (RR_BLOCK * RR_SQ_SHARDS));
    const uint64_t chunk = std::min<uint64_t>(s->tuning.shade_chunk_rays ? std::max<uint64_t>(65536, s->tuning.shade_chunk_rays) : (64ull << 20), chunk_by_lights);
    // level 1: fixed shadow slots, (enabled light, hit of the chunk), the chunk padded to whole workgroup iterations;
    // deeper levels: the dense sharded queue (a shard's static share of the chunk, one slack group per shard)
    const uint64_t sq_need = std::max<uint64_t>(1, (std::min<uint64_t>(chunk, std::max<uint64_t>(M, B)) + RR_BLOCK * RR_SQ_SHARDS) * std::max<uint32_t>(s->n_enabled_lights, 1u));
    if (sq_need > s->sq_cap) {
        for (int k = 0; k < 3; k++) HIP_TRY(s->sq[k].reserve(sq_need * 16));
        HIP_TRY(s->sq_valid.reserve((sq_need / RR_WAVE + 1) * 8));
        s->sq_cap = sq_need;
    }
    auto queue_at = [&](uint64_t base) {
        DRayQueue q;
        q.r0 = s->arena[0].as<float4>() + base; q.r1 = s->arena[1].as<float4>() + base;
        q.r2 = s->arena[2].as<uint2>() + base; q.hit = s->arena[3].as<uint4>() + base;
        return q;
    };
    DShadowQueue SQ{s->sq[0].as<float4>(), s->sq[1].as<float4>(), s->sq[2].as<float4>()};

    uint32_t* pool = s->pool.as<uint32_t>();
    unsigned long long* counters = s->counters.as<unsigned long long>();
    const int shadow_grid = s->n_cus * RR_SHADOW_WAVES; // RR_STACK_DEPTH KB of LDS stack per 256-thread workgroup
    const int shade_grid_max = s->n_cus * 2 * RR_SHADE_WAVES;
    const uint32_t L = s->n_enabled_lights;

    uint32_t next_word = 0;
    size_t pool_segment = 0; // 0 = s->pool, k = s->pool_more[k - 1]
    // Per-batch counters come out of zeroed segments of POOL_WORDS words.  A segment is never recycled inside a batch
    // (launches still in flight and the levels above in the recursion hold pointers into it); a batch with more
    // launches than one segment serves (a deeply branching scene in a very small ray arena) gets another one.
    auto words = [&](uint32_t n) -> uint32_t* {
        if (next_word + n > POOL_WORDS) {
            if (pool_segment == s->pool_more.size()) {
                if (s->pool_more.size() >= 255) return nullptr; // 4 GB of counters: something else is wrong
                s->pool_more.emplace_back();
                if (s->pool_more.back().reserve(POOL_WORDS * 4) != hipSuccess) { s->pool_more.pop_back(); return nullptr; }
            }
            pool = s->pool_more[pool_segment++].as<uint32_t>();
            if (hipMemsetAsync(pool, 0, POOL_WORDS * 4, st) != hipSuccess) return nullptr;
            next_word = 0;
        }
        uint32_t* p = pool + next_word; next_word += n; return p;
    };
    // One depth level: rays [base, base + n) of the arena, their count also in the device word `count`.
    // The size of the next level is read back once per slice (4 bytes + stream sync), so launches are sized by the
    // rays that exist and empty levels are never launched.
    // depth level 1 = the batch's primary rays [pr.first, pr.first + pr.n): only hit records (hit1); its children start the arena
    DPrimary pr{s->sample_xy.as<uint16_t>(), 0ull, 0u, 1u};
    std::function<int(uint32_t, uint64_t, uint64_t, uint32_t*)> run_level =
        [&](uint32_t d, uint64_t base, uint64_t n, uint32_t* count) -> int {
        DRayQueue qin = queue_at(base);
        if (d == 1) { qin.r0 = nullptr; qin.r1 = nullptr; qin.r2 = nullptr; qin.hit = s->hit1.as<uint4>(); }
        {
            uint32_t* head = words(1);
            if (!head) return fail(RR_ERR_UNSUPPORTED, "out of memory for the per-launch counters of a batch");
            ScopedTimer t(s, st, d == 1 ? 4 : 0);
            const int rcl = launch_trace_closest(s, d == 1, qin, count, head, n, s->shade_const.as<DShadeConst>(), s->region_xy.as<uint32_t>(), pr, counters, st);
            if (rcl != RR_OK) return rcl;
        }
        const bool spawns = d <= R; // the deepest level spawns nothing (k_shade: depth <= max_recursion)
        const uint64_t child_base = d == 1 ? 0 : base + n;
        // Children of a slice may use the space behind this level minus what the deeper levels need to make progress
        // themselves (one 256-ray slice = 512 children per spawning level below): the recursion can then never get stuck.
        uint64_t slice = n;
        if (spawns) {
            const uint64_t keep = 2ull * RR_BLOCK * (R - d); // spawning levels below d + 1's parent: d + 1 .. R
            const uint64_t room = M - child_base;
            if (room < keep + 2ull * RR_BLOCK) return fail(RR_ERR_OUT_OF_MEMORY, "ray arena of %llu rays is too small for depth level %u", (unsigned long long)M, d);
            if (2 * n > room - keep) { slice = ((room - keep) / 2 / RR_BLOCK) * RR_BLOCK; s->stats.sliced_levels++; }
        }
        for (uint64_t s0 = 0; s0 < n; s0 += slice) {
            const uint64_t s1 = std::min<uint64_t>(s0 + slice, n);
            uint32_t* child_count = words(1);
            if (!child_count) return fail(RR_ERR_UNSUPPORTED, "out of memory for the per-launch counters of a batch");
            const DRayQueue qout = queue_at(child_base);
            for (uint64_t c0 = s0; c0 < s1; c0 += chunk) {
                if (cancel && *cancel) { (void)hipStreamSynchronize(st); return fail(RR_ERR_CANCELLED, "cancelled"); }
                const uint64_t c1 = std::min<uint64_t>(c0 + chunk, s1);
                const int grid = (int)std::min<uint64_t>((c1 - c0 + RR_BLOCK - 1) / RR_BLOCK, (uint64_t)shade_grid_max);
                // level 1: shadow slots of this chunk = L x (the chunk padded to whole workgroup iterations), one validity word per 64
                // (only where the shadow kernel's packet form applies: rr_kernels.hip, RR_BEAM_MIN_ITEMS .. RR_BEAM_MAX_ITEMS)
                // and up to RR_FIXED_SLOT_LIGHTS enabled lights: k_shade keeps one bit per light and lane for the validity words; more lights
                // take the dense queue of the deeper levels, which has no such limit
                const bool sq_fixed = d == 1 && s->view.n_items >= RR_BEAM_MIN_ITEMS && s->view.n_items <= RR_BEAM_MAX_ITEMS && L <= RR_FIXED_SLOT_LIGHTS;
                const uint32_t sq_chunk_cap = sq_fixed ? (uint32_t)(((c1 - c0 + RR_BLOCK - 1) / RR_BLOCK) * RR_BLOCK) : 0u;
                unsigned long long* sq_valid = s->sq_valid.as<unsigned long long>();
                // deeper levels: shadow sub-queues, a shard gets the packets with (packet % RR_SQ_SHARDS == shard), L rays per hit at most
                const uint64_t groups = (c1 - c0 + RR_BLOCK - 1) / RR_BLOCK; // 256-ray groups, dealt round-robin to the shards
                const uint32_t segcap = (uint32_t)(((groups + RR_SQ_SHARDS - 1) / RR_SQ_SHARDS) * RR_BLOCK * std::max(L, 1u));
                next_word = (next_word + 31u) & ~31u; // the append counters start on a 128-B line
                uint32_t* sq_counts = words(RR_SQ_SHARDS * RR_SQ_STRIDE);
                uint32_t* shead = words(1);
                if (!sq_counts || !shead) return fail(RR_ERR_UNSUPPORTED, "out of memory for the per-launch counters of a batch");
                {
                    ScopedTimer t(s, st, d == 1 ? 5 : 2);
                    if (d == 1) hipLaunchKernelGGL(k_shade<true>, dim3(grid), dim3(RR_BLOCK), 0, st, s->shade_const.as<DShadeConst>(), s->region_xy.as<uint32_t>(), pr, qin, count,
                                                   (uint32_t)c0, (uint32_t)c1, qout, child_count, SQ, sq_counts, segcap, sq_valid, sq_chunk_cap, acc, counters);
                    else hipLaunchKernelGGL(k_shade<false>, dim3(grid), dim3(RR_BLOCK), 0, st, s->shade_const.as<DShadeConst>(), s->region_xy.as<uint32_t>(), pr, qin, count,
                                            (uint32_t)c0, (uint32_t)c1, qout, child_count, SQ, sq_counts, segcap, sq_valid, sq_chunk_cap, acc, counters);
                }
                // The size of the next level is final once the slice's last shade chunk has run: its read-back is enqueued
                // BEFORE that chunk's shadow kernel, so the host learns it (and enqueues the next level) while the shadow
                // rays are still being traced, instead of leaving the device idle for a host round trip per level.
                if (spawns && c1 == s1) {
                    HIP_TRY(hipMemcpyAsync(s->h_count, child_count, 4, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipEventRecord(s->count_ready, st));
                }
                if (L) {
                    ScopedTimer t(s, st, sq_fixed ? 6 : 1); // (by kernel BUILD: level 1 of a scene without fixed shadow slots runs k_trace_shadow<false>)
                    if (sq_fixed) {
                        const uint32_t sq_packets = (sq_chunk_cap / RR_WAVE) * L;
                        const int sgrid = (int)std::min<uint64_t>(((uint64_t)sq_packets * RR_WAVE + RR_BLOCK - 1) / RR_BLOCK, (uint64_t)shadow_grid);
                        hipLaunchKernelGGL(k_trace_shadow<true>, dim3(sgrid), dim3(RR_BLOCK), 0, st, s->view, SQ, sq_counts, segcap, sq_valid, sq_packets, shead, acc);
                    } else {
                        const uint64_t sq_ub = (c1 - c0) * L;
                        const int sgrid = (int)std::min<uint64_t>((sq_ub + RR_BLOCK - 1) / RR_BLOCK, (uint64_t)shadow_grid);
                        hipLaunchKernelGGL(k_trace_shadow<false>, dim3(sgrid), dim3(RR_BLOCK), 0, st, s->view, SQ, sq_counts, segcap, sq_valid, 0u, shead, acc);
                    }
                }
            }
            if (!spawns) continue;
            HIP_TRY(hipEventSynchronize(s->count_ready));
            const uint64_t m = *s->h_count;
            if (m > M - child_base) return fail(RR_ERR_DEVICE, "internal: level %u holds %llu rays, room for %llu", d + 1, (unsigned long long)m, (unsigned long long)(M - child_base));
            if (m == 0) continue;
            // On request (rr_tuning::bin_min_rays) deeper levels are traced in bins of (origin cell, direction octant) when
            // the sorted copy fits behind the unsorted one (rr_kernels.hip: ray binning; off by default, it does not pay).
            uint64_t level_base = child_base;
            const uint64_t bin_min = s->tuning.bin_min_rays;
            if (bin_min != 0 && m >= bin_min && M - child_base >= 3 * m + 2ull * RR_BLOCK * (R + 1)) {
                next_word = (next_word + 31u) & ~31u;
                int* bounds = (int*)words(8);
                uint32_t* hist = words(RR_BIN_COUNT);
                if (bounds && hist) {
                    const int init[8] = {0x7f7fffff, 0x7f7fffff, 0x7f7fffff, (int)0x80800000, (int)0x80800000, (int)0x80800000, 0, 0}; // ordered(+FLT_MAX) x3, ordered(-FLT_MAX) x3
                    HIP_TRY(hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, st));
                    const DRayQueue qsrc = queue_at(child_base), qdst = queue_at(child_base + m);
                    const int g = (int)std::min<uint64_t>((m + RR_BLOCK - 1) / RR_BLOCK, (uint64_t)s->n_cus * 8);
                    ScopedTimer t(s, st, 3);
                    hipLaunchKernelGGL(k_bin_bounds, dim3(g), dim3(RR_BLOCK), 0, st, qsrc, (uint32_t)m, bounds);
                    hipLaunchKernelGGL(k_bin_count, dim3(g), dim3(RR_BLOCK), 0, st, qsrc, (uint32_t)m, bounds, hist);
                    hipLaunchKernelGGL(k_bin_prefix, dim3(1), dim3(1024), 0, st, hist);
                    hipLaunchKernelGGL(k_bin_scatter, dim3(g), dim3(RR_BLOCK), 0, st, qsrc, qdst, (uint32_t)m, hist);
                    level_base = child_base + m;
                    s->stats.binned_rays += m;
                }
            }
            { const int rc2 = run_level(d + 1, level_base, m, child_count); if (rc2 != RR_OK) return rc2; }
        }
        return RR_OK;
    };

    HIP_TRY(hipEventRecord(s->frame_a, st));
    for (uint64_t first = 0; first < total_primary; first += B) {
        if (cancel && *cancel) { (void)hipStreamSynchronize(st); return fail(RR_ERR_CANCELLED, "cancelled"); }
        const uint32_t n_batch = (uint32_t)std::min<uint64_t>(B, total_primary - first);
        pool = s->pool.as<uint32_t>(); pool_segment = 0;
        HIP_TRY(hipMemsetAsync(pool, 0, POOL_WORDS * 4, st));
        next_word = 0;
        uint32_t* level1_count = words(1);
        // The batch covers primary indices [first, first + n_batch): index i -> sample i / npix, pixel i % npix.
        pr.first = first; pr.n = n_batch;
        pr.group = (n_batch % ((uint64_t)npix * G) == 0 && first % npix == 0) ? G : 1u;
        s->stats.batches++;
        const int rcl = run_level(1, 0, n_batch, level1_count);
        if (rcl != RR_OK) return rcl;
        HIP_TRY(hipGetLastError());
        // batches are stream-ordered; only a caller that can cancel needs the host to keep pace with the device
        if (cancel && first + B < total_primary) HIP_TRY(hipStreamSynchronize(st));
        const uint64_t done = first + n_batch;
        if (hook && hook->fn && done < total_primary && done % npix == 0) {
            DFrame pf = fr;
            pf.samples = (uint32_t)(done / npix); // the mean over the sample slices finished so far
            hipLaunchKernelGGL(k_resolve, dim3((npix + RR_BLOCK - 1) / RR_BLOCK), dim3(RR_BLOCK), 0, st, pf, s->region_xy.as<uint32_t>(), s->trace_order.as<uint32_t>(), acc,
                               out->rgba8, out->normal, out->depth, out->object_id, frame_layout ? 1u : 0u);
            void* const dev[4] = {out->rgba8, out->normal, out->depth, out->object_id};
            for (int k = 0; k < 4; k++)
                if (hook->host[k] && dev[k]) HIP_TRY(hipMemcpyAsync(hook->host[k], dev[k], hook->bytes[k], hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (hook->fn(hook->user, done, total_primary) != 0) return fail(RR_ERR_CANCELLED, "stopped by the pass callback");
        }
    }
    hipLaunchKernelGGL(k_resolve, dim3((npix + RR_BLOCK - 1) / RR_BLOCK), dim3(RR_BLOCK), 0, st, fr, s->region_xy.as<uint32_t>(), s->trace_order.as<uint32_t>(), acc,
                       out->rgba8, out->normal, out->depth, out->object_id, frame_layout ? 1u : 0u);
    HIP_TRY(hipEventRecord(s->frame_b, st));
    HIP_TRY(hipGetLastError());
    // a scene that branches more than the arena was sized for gets a larger one for its next frame (within the budget)
    if (s->stats.sliced_levels > 0 && s->arena_factor < 128) s->arena_factor *= 2;
    return RR_OK;
}

extern "C" int rr_render_region_device(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy,
                                       const rr_region* rg, const rr_frame* out, void* hip_stream, const volatile int* cancel) try {
    int rc = check_frame_args(s, cam, cfg, sample_xy);
    if (rc != RR_OK) return rc;
    rc = check_region(cam->width, cam->height, rg);
    if (rc != RR_OK) return rc;
    if (!out || !out->rgba8) return fail(RR_ERR_INVALID_ARGUMENT, "out->rgba8 is required");
    std::lock_guard<std::mutex> lk(s->mu);
    return render_region_locked(s, cam, cfg, sample_xy, rg, out, false, (hipStream_t)hip_stream, cancel);
} RR_GUARD_END("rr_render_region_device")

static int render_to_host(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy, const rr_frame* out,
                          const volatile int* cancel, rr_pass_fn fn, void* user, uint32_t min_passes) {
    int rc = check_frame_args(s, cam, cfg, sample_xy);
    if (rc != RR_OK) return rc;
    if (!out || !out->rgba8) return fail(RR_ERR_INVALID_ARGUMENT, "out->rgba8 is required");
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    const size_t np = (size_t)cam->width * cam->height;
    const size_t bytes[4] = {np * 4, np * 12, np * 4, np * 4};
    void* host[4] = {out->rgba8, out->normal, out->depth, out->object_id};
    rr_frame dev{};
    void** devp[4] = {(void**)&dev.rgba8, (void**)&dev.normal, (void**)&dev.depth, (void**)&dev.object_id};
    for (int k = 0; k < 4; k++)
        if (host[k]) { HIP_TRY(s->tmp_out[k].reserve(bytes[k])); *devp[k] = s->tmp_out[k].p; }
    rr_region whole{8, 8, 1, 0}; // 8x8 tiles: one wave = one tile of primary rays
    PassHook hook{fn, user, min_passes, {host[0], host[1], host[2], host[3]}, {bytes[0], bytes[1], bytes[2], bytes[3]}};
    rc = render_region_locked(s, cam, cfg, sample_xy, &whole, &dev, true, nullptr, cancel, fn ? &hook : nullptr);
    if (rc != RR_OK) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    for (int k = 0; k < 4; k++)
        if (host[k]) HIP_TRY(hipMemcpy(host[k], s->tmp_out[k].p, bytes[k], hipMemcpyDeviceToHost));
    return RR_OK;
}

extern "C" int rr_render(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy, const rr_frame* out,
                         const volatile int* cancel) try {
    return render_to_host(s, cam, cfg, sample_xy, out, cancel, nullptr, nullptr, 0);
} RR_GUARD_END("rr_render")

extern "C" int rr_render_progressive(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy, const rr_frame* out,
                                     uint32_t min_passes, rr_pass_fn on_pass, void* user, const volatile int* cancel) try {
    if (!on_pass) return fail(RR_ERR_INVALID_ARGUMENT, "on_pass is required (use rr_render for a one-shot frame)");
    return render_to_host(s, cam, cfg, sample_xy, out, cancel, on_pass, user, min_passes);
} RR_GUARD_END("rr_render_progressive")

static int collect_stats_locked(rr_scene* s);
extern "C" int rr_scene_last_stats(const rr_scene* cs, rr_frame_stats* out) try {
    if (!cs || !out) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    rr_scene* s = const_cast<rr_scene*>(cs);
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    if (!s->stats_final) { const int rc = collect_stats_locked(s); if (rc != RR_OK) return rc; }
    *out = s->stats;
    return RR_OK;
} RR_GUARD_END("rr_scene_last_stats")
// the device counters and launch timers of the frame (or pass) that ran last, into s->stats
static int collect_stats_locked(rr_scene* s) {
    float ms = 0.0f;
    if (hipEventSynchronize(s->frame_b) == hipSuccess && hipEventElapsedTime(&ms, s->frame_a, s->frame_b) == hipSuccess) s->stats.ms_total = ms;
    resolve_timers(s);
    unsigned long long c[RR_CNT_WORDS];
    HIP_TRY(hipMemcpy(c, s->counters.p, sizeof c, hipMemcpyDeviceToHost));
    s->stats.primary_rays = c[RR_CNT_PRIMARY]; s->stats.secondary_rays = c[RR_CNT_SECONDARY];
    s->stats.shadow_rays = c[RR_CNT_SHADOW]; s->stats.shaded_hits = c[RR_CNT_SHADED];
    return RR_OK;
}

// The frame filled in TILE BY TILE, every pixel final when it appears: what the reference's GUI shows (shuffled 2x2 cells, each rendered with all of
// its samples: src/renderer.rs:125-172, drained by Run::apply_pixels, src/run.rs:506-545).  Pass k of n_passes renders the 32x8-pixel tiles with
// tile_index % n_passes == k -- an interleaved subset, like the shuffled cell list -- straight into their places in the frame.
extern "C" int rr_render_progressive_tiles(rr_scene* s, const rr_camera* cam, const rr_config* cfg, const uint16_t* sample_xy, const rr_frame* out,
                                           uint32_t n_passes, rr_pass_fn on_pass, void* user, const volatile int* cancel) try {
    if (!on_pass) return fail(RR_ERR_INVALID_ARGUMENT, "on_pass is required (use rr_render for a one-shot frame)");
    int rc = check_frame_args(s, cam, cfg, sample_xy);
    if (rc != RR_OK) return rc;
    if (!out || !out->rgba8) return fail(RR_ERR_INVALID_ARGUMENT, "out->rgba8 is required");
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t W = cam->width, H = cam->height, TW = 32, TH = 8;
    const size_t np = (size_t)W * H;
    const size_t bytes[4] = {np * 4, np * 12, np * 4, np * 4};
    void* host[4] = {out->rgba8, out->normal, out->depth, out->object_id};
    rr_frame dev{};
    void** devp[4] = {(void**)&dev.rgba8, (void**)&dev.normal, (void**)&dev.depth, (void**)&dev.object_id};
    for (int k = 0; k < 4; k++)
        if (host[k]) { HIP_TRY(s->tmp_out[k].reserve(bytes[k])); *devp[k] = s->tmp_out[k].p; HIP_TRY(hipMemsetAsync(s->tmp_out[k].p, 0, bytes[k], nullptr)); }
    const uint32_t n_tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    const uint32_t P = std::max(1u, std::min(n_passes ? n_passes : 16u, n_tiles));
    rr_frame_stats sum{};
    uint64_t done = 0;
    for (uint32_t k = 0; k < P; k++) {
        if (cancel && *cancel) return fail(RR_ERR_CANCELLED, "cancelled");
        const rr_region rg{TW, TH, P, k};
        rc = render_region_locked(s, cam, cfg, sample_xy, &rg, &dev, true, nullptr, cancel);
        if (rc != RR_OK) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
        rc = collect_stats_locked(s);
        if (rc != RR_OK) return rc;
        {   // the frame's statistics are the sums over its passes
            const rr_frame_stats& a = s->stats;
            sum.primary_rays += a.primary_rays; sum.secondary_rays += a.secondary_rays; sum.shadow_rays += a.shadow_rays; sum.shaded_hits += a.shaded_hits;
            sum.ms_total += a.ms_total; sum.ms_trace_closest += a.ms_trace_closest; sum.ms_trace_shadow += a.ms_trace_shadow; sum.ms_shade += a.ms_shade;
            sum.launches_trace_closest += a.launches_trace_closest; sum.launches_trace_shadow += a.launches_trace_shadow; sum.launches_shade += a.launches_shade;
            sum.batches += a.batches; sum.sliced_levels += a.sliced_levels; sum.binned_rays += a.binned_rays; sum.ms_binning += a.ms_binning;
            sum.ms_trace_closest_level1 += a.ms_trace_closest_level1; sum.launches_trace_closest_level1 += a.launches_trace_closest_level1;
            sum.ms_shade_level1 += a.ms_shade_level1; sum.launches_shade_level1 += a.launches_shade_level1;
            sum.ms_trace_shadow_level1 += a.ms_trace_shadow_level1; sum.launches_trace_shadow_level1 += a.launches_trace_shadow_level1;
        }
        for (int b = 0; b < 4; b++)
            if (host[b]) HIP_TRY(hipMemcpy(host[b], s->tmp_out[b].p, bytes[b], hipMemcpyDeviceToHost));
        done += rr_region_pixel_count(W, H, &rg);
        s->stats = sum; s->stats_final = true;
        if (k + 1 < P && on_pass(user, done * cfg->samples, (uint64_t)np * cfg->samples) != 0) return fail(RR_ERR_CANCELLED, "stopped by the pass callback");
    }
    return RR_OK;
} RR_GUARD_END("rr_render_progressive_tiles")

extern "C" int rr_scene_set_compat(rr_scene* s, uint32_t flags) try {
    if (!s) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (flags & ~RR_COMPAT_OCCLUDER_ALPHA_SHADOWS) return fail(RR_ERR_INVALID_ARGUMENT, "unknown compatibility flags 0x%x", flags);
    std::lock_guard<std::mutex> lk(s->mu);
    s->view.compat = (s->view.compat & RR_VIEW_NAN_BALLS) | flags; // the scene view is passed to the kernels by value with every launch
    return RR_OK;
} RR_GUARD_END("rr_scene_set_compat")

extern "C" int rr_scene_set_tuning(rr_scene* s, const rr_tuning* t) try {
    if (!s || !t) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (t->struct_size != sizeof(rr_tuning)) return fail(RR_ERR_INVALID_ARGUMENT, "rr_tuning::struct_size %u, library expects %zu", t->struct_size, sizeof(rr_tuning));
    if (t->sample_group > 64u || (t->sample_group & (t->sample_group - 1u))) return fail(RR_ERR_INVALID_ARGUMENT, "sample_group %u is not 0 or a power of two <= 64", t->sample_group);
    std::lock_guard<std::mutex> lk(s->mu);
    s->tuning = *t;
    s->profiling = t->kernel_timing != 0;
    return RR_OK;
} RR_GUARD_END("rr_scene_set_tuning")
extern "C" int rr_scene_get_tuning(const rr_scene* s, rr_tuning* t) try {
    if (!s || !t) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    *t = s->tuning;
    t->struct_size = (uint32_t)sizeof(rr_tuning);
    return RR_OK;
} RR_GUARD_END("rr_scene_get_tuning")

// ---------------------------------------------------------------------------
// multi-GPU epilogue: compact per-rank buffers (concatenated in rank order) -> frame order
// ---------------------------------------------------------------------------
struct GatherMap { DevBuf index; uint32_t w, h, tw, th, n; int device; };
static std::mutex g_gather_mu;
static std::vector<GatherMap*> g_gather_maps;

extern "C" int rr_deinterleave_device(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks,
                                      uint32_t elem_bytes, const void* src, void* dst, int device, void* hip_stream) try {
    rr_region probe{tile_w, tile_h, n_ranks, 0};
    int rc = check_region(width, height, &probe);
    if (rc != RR_OK) return rc;
    if (!src || !dst || elem_bytes == 0 || (elem_bytes & 3u)) return fail(RR_ERR_INVALID_ARGUMENT, "bad buffers or elem_bytes %u", elem_bytes);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_gather_mu);
    GatherMap* gm = nullptr;
    for (GatherMap* m : g_gather_maps)
        if (m->w == width && m->h == height && m->tw == tile_w && m->th == tile_h && m->n == n_ranks && m->device == device) gm = m;
    const uint32_t np = width * height;
    if (!gm) {
        std::vector<uint32_t> index(np), xy;
        uint32_t base = 0;
        for (uint32_t r = 0; r < n_ranks; r++) {
            rr_region rg{tile_w, tile_h, n_ranks, r};
            fill_region(width, height, rg, &xy);
            for (uint32_t p = 0; p < xy.size(); p++) index[(size_t)(xy[p] >> 16) * width + (xy[p] & 0xffffu)] = base + p;
            base += (uint32_t)xy.size();
        }
        gm = new GatherMap{DevBuf(), width, height, tile_w, tile_h, n_ranks, device};
        HIP_TRY(gm->index.reserve((size_t)np * 4));
        HIP_TRY(hipMemcpy(gm->index.p, index.data(), (size_t)np * 4, hipMemcpyHostToDevice));
        g_gather_maps.push_back(gm);
    }
    const uint32_t words = elem_bytes / 4;
    const uint64_t total = (uint64_t)np * words;
    hipLaunchKernelGGL(k_gather_frame, dim3((uint32_t)((total + RR_BLOCK - 1) / RR_BLOCK)), dim3(RR_BLOCK), 0, (hipStream_t)hip_stream,
                       gm->index.as<uint32_t>(), np, words, (const uint32_t*)src, (uint32_t*)dst);
    HIP_TRY(hipGetLastError());
    return RR_OK;
} RR_GUARD_END("rr_deinterleave_device")

// The gathered packs of a multi-rank frame -> the four frame-order buffers, one launch (k_gather_packed).
struct GatherMap2 { DevBuf rank, local; uint32_t w, h, tw, th, n; int device; };
static std::vector<GatherMap2*> g_gather_maps2;
extern "C" int rr_deinterleave_packed_device(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks,
                                             const void* packs, uint64_t pack_stride, const uint64_t* section_offset, const uint32_t* elem_bytes,
                                             void* const* dst, int device, void* hip_stream) try {
    rr_region probe{tile_w, tile_h, n_ranks, 0};
    int rc = check_region(width, height, &probe);
    if (rc != RR_OK) return rc;
    if (!packs || !section_offset || !elem_bytes || !dst) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    DPackedGather g{};
    for (int k = 0; k < 4; k++) {
        if (elem_bytes[k] & 3u) return fail(RR_ERR_INVALID_ARGUMENT, "elem_bytes[%d] = %u is not a multiple of 4", k, elem_bytes[k]);
        if (elem_bytes[k] && !dst[k]) return fail(RR_ERR_INVALID_ARGUMENT, "dst[%d] is NULL for a present buffer", k);
        if ((section_offset[k] & 3u) || (pack_stride & 3u)) return fail(RR_ERR_INVALID_ARGUMENT, "sections and packs must be 4-byte aligned");
        g.words[k] = elem_bytes[k] / 4u; g.words_total += g.words[k]; g.section[k] = section_offset[k]; g.dst[k] = (uint32_t*)dst[k];
    }
    if (g.words_total == 0) return fail(RR_ERR_INVALID_ARGUMENT, "no buffer to move");
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_gather_mu);
    GatherMap2* gm = nullptr;
    for (GatherMap2* m : g_gather_maps2)
        if (m->w == width && m->h == height && m->tw == tile_w && m->th == tile_h && m->n == n_ranks && m->device == device) gm = m;
    const uint32_t np = width * height;
    if (!gm) {
        std::vector<uint32_t> rank(np), local(np), xy;
        for (uint32_t r = 0; r < n_ranks; r++) {
            rr_region rg{tile_w, tile_h, n_ranks, r};
            fill_region(width, height, rg, &xy);
            for (uint32_t p = 0; p < xy.size(); p++) { const size_t o = (size_t)(xy[p] >> 16) * width + (xy[p] & 0xffffu); rank[o] = r; local[o] = p; }
        }
        gm = new GatherMap2{DevBuf(), DevBuf(), width, height, tile_w, tile_h, n_ranks, device};
        HIP_TRY(gm->rank.reserve((size_t)np * 4)); HIP_TRY(gm->local.reserve((size_t)np * 4));
        HIP_TRY(hipMemcpy(gm->rank.p, rank.data(), (size_t)np * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(gm->local.p, local.data(), (size_t)np * 4, hipMemcpyHostToDevice));
        g_gather_maps2.push_back(gm);
    }
    g.src_rank = gm->rank.as<uint32_t>(); g.src_local = gm->local.as<uint32_t>();
    g.packs = (const char*)packs; g.pack_stride = pack_stride; g.n_pixels = np;
    const uint64_t total = (uint64_t)np * g.words_total;
    hipLaunchKernelGGL(k_gather_packed, dim3((uint32_t)((total + RR_BLOCK - 1) / RR_BLOCK)), dim3(RR_BLOCK), 0, (hipStream_t)hip_stream, g);
    HIP_TRY(hipGetLastError());
    return RR_OK;
} RR_GUARD_END("rr_deinterleave_packed_device")

// Lock order of a set of scene handles: by address (std::less is a total order on pointers).
static std::vector<rr_scene*> multi_lock_order(rr_scene* const* scenes, uint32_t n) {
    std::vector<rr_scene*> v(scenes, scenes + n);
    std::sort(v.begin(), v.end(), std::less<rr_scene*>());
    return v;
}
// test hook (tests/test_abi.py): the order in which rr_render_multi would lock `scenes`, as indices into the caller's array
extern "C" int rr_multi_lock_order(rr_scene* const* scenes, uint32_t n_scenes, uint32_t* order_out) try {
    if (!scenes || !order_out || n_scenes == 0) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    const std::vector<rr_scene*> v = multi_lock_order(scenes, n_scenes);
    for (uint32_t k = 0; k < n_scenes; k++)
        for (uint32_t i = 0; i < n_scenes; i++) if (scenes[i] == v[k]) { order_out[k] = i; break; }
    return RR_OK;
} RR_GUARD_END("rr_multi_lock_order")

// Peer access between two devices, both ways: checked once per ordered pair, enabled on first use.
// false = no direct path (the caller stages through the host).  The same device counts as direct.
static std::mutex g_peer_mu;
static std::map<std::pair<int, int>, bool> g_peer_state; // (from, to) -> `from` may access memory of `to`
static bool enable_peer_one_way(int from, int to) {
    auto it = g_peer_state.find({from, to});
    if (it != g_peer_state.end()) return it->second;
    bool ok = false;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (hipSetDevice(from) == hipSuccess) {
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            (void)hipGetLastError(); // "already enabled" is not an error of this call
        }
        (void)hipSetDevice(cur);
    }
    g_peer_state[{from, to}] = ok;
    return ok;
}
static bool ensure_peer_access(int a, int b) {
    if (a == b) return true;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    const bool ab = enable_peer_one_way(a, b), ba = enable_peer_one_way(b, a);
    return ab && ba;
}

// ---------------------------------------------------------------------------
// one frame on several GPUs from ONE host process (the reference host is one process, src/renderer.rs:105-172):
// one host thread per device renders that device's interleaved tiles, the compact per-device buffers are copied
// peer-to-peer (xGMI) into device 0, de-interleaved there and copied to the host once.  No collective library is
// involved: the exchange is n - 1 point-to-point copies of 1 / n of the frame each.  Every device works on its own
// non-blocking stream.  UNVERIFIED ON N > 1 DEVICES until an N-GPU node has run it (the pool hands out 1-GPU boxes;
// tests/test_gpu_multi.py puts several handles on device 0).
// ---------------------------------------------------------------------------
extern "C" int rr_render_multi(rr_scene* const* scenes, uint32_t n_scenes, const rr_camera* cam, const rr_config* cfg,
                               const uint16_t* sample_xy, const rr_frame* out, const volatile int* cancel) try {
    if (!scenes || n_scenes == 0) return fail(RR_ERR_INVALID_ARGUMENT, "no scenes");
    if (n_scenes > 64) return fail(RR_ERR_UNSUPPORTED, "%u scene handles", n_scenes);
    for (uint32_t i = 0; i < n_scenes; i++) {
        if (!scenes[i]) return fail(RR_ERR_INVALID_ARGUMENT, "scene %u is NULL", i);
        for (uint32_t j = 0; j < i; j++) if (scenes[j] == scenes[i]) return fail(RR_ERR_INVALID_ARGUMENT, "scene handle %u is passed twice", i);
        int rc = check_frame_args(scenes[i], cam, cfg, sample_xy);
        if (rc != RR_OK) return rc;
    }
    if (!out || !out->rgba8) return fail(RR_ERR_INVALID_ARGUMENT, "out->rgba8 is required");
    const uint32_t W = cam->width, H = cam->height, TW = 32, TH = 8; // interleaved 32x8 tiles: tile_index % n == device slot
    const size_t np = (size_t)W * H;
    const size_t esz[4] = {4, 12, 4, 4};
    void* host[4] = {out->rgba8, out->normal, out->depth, out->object_id};
    std::vector<uint64_t> count(n_scenes), offset(n_scenes);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_scenes; i++) {
        rr_region rg{TW, TH, n_scenes, i};
        count[i] = rr_region_pixel_count(W, H, &rg);
        offset[i] = total; total += count[i];
    }
    rr_scene* s0 = scenes[0];
    // Handles are locked in ADDRESS order, whatever order the caller passed them in: two calls that share handles in
    // opposite orders (or a call racing rr_render on one of them) then serialise instead of deadlocking.
    std::vector<std::unique_lock<std::mutex>> locks;
    for (rr_scene* s : multi_lock_order(scenes, n_scenes)) locks.emplace_back(s->mu);
    // Peer access between device 0 and every other device taking part: checked, and enabled both ways on first use.
    // A pair without it does not fall back silently to whatever hipMemcpyPeerAsync does: its buffers are staged
    // through pinned host memory here, and the frame's stats say so.
    std::vector<char> direct(n_scenes, 1);
    uint32_t n_peer = 0, n_staged = 0;
    for (uint32_t i = 1; i < n_scenes; i++) {
        direct[i] = (s0->tuning.multi_force_staged == 0u && ensure_peer_access(scenes[i]->device, s0->device)) ? 1 : 0;
        if (direct[i]) n_peer++; else n_staged++;
    }
    auto own_stream = [](rr_scene* s) -> int { // on the scene's device
        if (!s->multi_stream) HIP_TRY(hipStreamCreateWithFlags(&s->multi_stream, hipStreamNonBlocking));
        return RR_OK;
    };
    // device 0: the concatenation of the compact buffers (rank order) and the frame-order buffers
    HIP_TRY(hipSetDevice(s0->device));
    { int rc = own_stream(s0); if (rc != RR_OK) return rc; }
    for (int k = 0; k < 4; k++)
        if (host[k]) { HIP_TRY(s0->multi_cat[k].reserve(np * esz[k])); HIP_TRY(s0->tmp_out[k].reserve(np * esz[k])); }
    // every device renders its tiles into its own compact buffers on its own stream, then pushes them towards device 0
    std::vector<int> rcs(n_scenes, RR_OK);
    std::vector<std::string> errs(n_scenes);
    auto work = [&](uint32_t i) {
        rr_scene* s = scenes[i];
        auto body = [&]() -> int {
            HIP_TRY(hipSetDevice(s->device));
            { int rc = own_stream(s); if (rc != RR_OK) return rc; }
            rr_frame dev{};
            void** devp[4] = {(void**)&dev.rgba8, (void**)&dev.normal, (void**)&dev.depth, (void**)&dev.object_id};
            for (int k = 0; k < 4; k++) {
                if (!host[k]) continue;
                if (i == 0) *devp[k] = (char*)s0->multi_cat[k].p + offset[0] * esz[k]; // device 0 renders straight into its slot
                else { HIP_TRY(s->multi_part[k].reserve(std::max<uint64_t>(count[i], 1) * esz[k])); *devp[k] = s->multi_part[k].p; }
            }
            rr_region rg{TW, TH, n_scenes, i};
            int rc = render_region_locked(s, cam, cfg, sample_xy, &rg, &dev, false, s->multi_stream, cancel);
            if (rc != RR_OK) return rc;
            if (i != 0)
                for (int k = 0; k < 4; k++) {
                    if (!host[k] || !count[i]) continue;
                    const size_t bytes = count[i] * esz[k];
                    void* dst = (char*)s0->multi_cat[k].p + offset[i] * esz[k];
                    if (direct[i] && s->device == s0->device) HIP_TRY(hipMemcpyAsync(dst, s->multi_part[k].p, bytes, hipMemcpyDeviceToDevice, s->multi_stream));
                    else if (direct[i]) HIP_TRY(hipMemcpyPeerAsync(dst, s0->device, s->multi_part[k].p, s->device, bytes, s->multi_stream));
                    else { // no peer access: device -> pinned host here, host -> device 0 after the join
                        if (s->multi_stage_bytes[k] < bytes) {
                            if (s->multi_stage[k]) { (void)hipHostFree(s->multi_stage[k]); s->multi_stage[k] = nullptr; s->multi_stage_bytes[k] = 0; }
                            HIP_TRY(hipHostMalloc(&s->multi_stage[k], bytes, hipHostMallocPortable));
                            s->multi_stage_bytes[k] = bytes;
                        }
                        HIP_TRY(hipMemcpyAsync(s->multi_stage[k], s->multi_part[k].p, bytes, hipMemcpyDeviceToHost, s->multi_stream));
                    }
                }
            HIP_TRY(hipStreamSynchronize(s->multi_stream));
            return RR_OK;
        };
        try { RR_FAULT_POINT("render_multi.worker"); rcs[i] = body(); }
        catch (...) { rcs[i] = guard_fail("rr_render_multi (device worker)"); }
        if (rcs[i] != RR_OK) { try { errs[i] = tl_error; } catch (...) { } } // the message lives in the worker's thread-local slot
    };
    {
        Workers threads; // joined on every path out of this block
        std::vector<char> inline_run(n_scenes, 0);
        for (uint32_t i = 1; i < n_scenes; i++)
            if (!threads.spawn([&work, i]() { work(i); })) inline_run[i] = 1;
        work(0);
        for (uint32_t i = 1; i < n_scenes; i++) if (inline_run[i]) work(i); // a thread that could not be started: its device waits for ours
        threads.join_and_rethrow();
    }
    const auto t_joined = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < n_scenes; i++)
        if (rcs[i] != RR_OK) return fail(rcs[i], "device slot %u: %s", i, errs[i].c_str());
    HIP_TRY(hipSetDevice(s0->device));
    for (uint32_t i = 1; i < n_scenes; i++) {
        if (direct[i]) continue;
        for (int k = 0; k < 4; k++)
            if (host[k] && count[i])
                HIP_TRY(hipMemcpyAsync((char*)s0->multi_cat[k].p + offset[i] * esz[k], scenes[i]->multi_stage[k], count[i] * esz[k], hipMemcpyHostToDevice, s0->multi_stream));
    }
    for (int k = 0; k < 4; k++) {
        if (!host[k]) continue;
        int rc = rr_deinterleave_device(W, H, TW, TH, n_scenes, (uint32_t)esz[k], s0->multi_cat[k].p, s0->tmp_out[k].p, s0->device, s0->multi_stream);
        if (rc != RR_OK) return rc;
    }
    for (int k = 0; k < 4; k++)
        if (host[k]) HIP_TRY(hipMemcpyAsync(host[k], s0->tmp_out[k].p, np * esz[k], hipMemcpyDeviceToHost, s0->multi_stream));
    HIP_TRY(hipStreamSynchronize(s0->multi_stream));
    s0->stats.multi_devices = n_scenes; s0->stats.multi_peer_links = n_peer; s0->stats.multi_staged_links = n_staged;
    s0->stats.ms_multi_exchange = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_joined).count();
    return RR_OK;
} RR_GUARD_END("rr_render_multi")

// ---------------------------------------------------------------------------
// post-processing (reference src/post_processing.rs:123-181)
// ---------------------------------------------------------------------------
extern "C" int rr_post_process_device(uint32_t width, uint32_t height, int cavity, int outline, const uint8_t* rgba_in,
                                      const float* normal, const uint32_t* object_id, uint8_t* rgba_out, int device, void* hip_stream) try {
    if (width == 0 || height == 0) return fail(RR_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", width, height);
    if (!rgba_in || !rgba_out || rgba_in == rgba_out) return fail(RR_ERR_INVALID_ARGUMENT, "rgba_in / rgba_out must be distinct non-NULL buffers");
    if ((cavity && !normal) || (outline && !object_id)) return fail(RR_ERR_INVALID_ARGUMENT, "cavity needs the normal buffer, outline the object-id buffer");
    HIP_TRY(hipSetDevice(device));
    const uint64_t n = (uint64_t)width * height;
    hipLaunchKernelGGL(k_post_process, dim3((uint32_t)((n + RR_BLOCK - 1) / RR_BLOCK)), dim3(RR_BLOCK), 0, (hipStream_t)hip_stream, width, height,
                       cavity ? 1u : 0u, outline ? 1u : 0u, (const uint32_t*)rgba_in, normal, object_id, (uint32_t*)rgba_out);
    HIP_TRY(hipGetLastError());
    return RR_OK;
} RR_GUARD_END("rr_post_process_device")

extern "C" int rr_post_process(uint32_t width, uint32_t height, int cavity, int outline, const uint8_t* rgba_in, const float* normal,
                               const uint32_t* object_id, uint8_t* rgba_out, int device) try {
    if (width == 0 || height == 0) return fail(RR_ERR_INVALID_ARGUMENT, "bad frame size %ux%u", width, height);
    if (!rgba_in || !rgba_out) return fail(RR_ERR_INVALID_ARGUMENT, "NULL image");
    if ((cavity && !normal) || (outline && !object_id)) return fail(RR_ERR_INVALID_ARGUMENT, "cavity needs the normal buffer, outline the object-id buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RR_ERR_NO_DEVICE, "no HIP device available");
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)width * height;
    DevBuf in, out, nrm, ids;
    HIP_TRY(in.reserve(n * 4)); HIP_TRY(out.reserve(n * 4));
    HIP_TRY(hipMemcpy(in.p, rgba_in, n * 4, hipMemcpyHostToDevice));
    if (normal) { HIP_TRY(nrm.reserve(n * 12)); HIP_TRY(hipMemcpy(nrm.p, normal, n * 12, hipMemcpyHostToDevice)); }
    if (object_id) { HIP_TRY(ids.reserve(n * 4)); HIP_TRY(hipMemcpy(ids.p, object_id, n * 4, hipMemcpyHostToDevice)); }
    int rc = rr_post_process_device(width, height, cavity, outline, in.as<uint8_t>(), nrm.as<float>(), ids.as<uint32_t>(), out.as<uint8_t>(), device, nullptr);
    if (rc == RR_OK) {
        hipError_t e = hipMemcpy(rgba_out, out.p, n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RR_ERR_DEVICE, "copy back: %s", hipGetErrorString(e));
    }
    in.release(); out.release(); nrm.release(); ids.release();
    return rc;
} RR_GUARD_END("rr_post_process")

// ---------------------------------------------------------------------------
// pick (reference src/raytracing.rs:237-273): pixel-centre ray, one closest-hit query
// ---------------------------------------------------------------------------
extern "C" int rr_pick(rr_scene* s, const rr_camera* cam, int x, int y, rr_pick_result* out) try {
    if (!s || !cam || !out) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (x < 0 || y < 0 || (uint32_t)x >= cam->width || (uint32_t)y >= cam->height) return fail(RR_ERR_INVALID_ARGUMENT, "pixel (%d,%d) outside %ux%u", x, y, cam->width, cam->height);
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    {
        double need[3];
        camera_reach(cam, nullptr, need);
        int rc = ensure_tlas_reach(s, need);
        if (rc != RR_OK) return rc;
    }
    DFrame fr;
    memset(&fr, 0, sizeof fr);
    memcpy(fr.proj_inv, cam->projection_inverse, 64);
    memcpy(fr.view_inv, cam->view_inverse, 64);
    fr.width = cam->width; fr.height = cam->height; fr.samples = 1; fr.cell_size = 1; fr.n_region_pixels = 1;
    DevBuf scratch;
    HIP_TRY(scratch.reserve(256 + sizeof(DShadeConst)));
    // layout: [0] region_xy, [4] sample_xy (2 x u16), [64] hit, [96] count, [100] head, [128] counters, [256] scene view + frame constants
    char* b = scratch.as<char>();
    uint32_t h_xy = (uint32_t)x | ((uint32_t)y << 16);
    HIP_TRY(hipMemset(b, 0, 256));
    HIP_TRY(hipMemcpy(b, &h_xy, 4, hipMemcpyHostToDevice));
    DRayQueue q{nullptr, nullptr, nullptr, (uint4*)(b + 64)};
    DPrimary pr{(const uint16_t*)(b + 4), 0ull, 1u, 1u};
    DShadeConst hc;
    hc.sc = s->view; hc.fr = fr;
    HIP_TRY(hipMemcpy(b + 256, &hc, sizeof hc, hipMemcpyHostToDevice));
    { const int rc = launch_trace_closest(s, true, q, (uint32_t*)(b + 96), (uint32_t*)(b + 100), 1, (const DShadeConst*)(b + 256), (const uint32_t*)b, pr, (unsigned long long*)(b + 128), nullptr);
      if (rc != RR_OK) return rc; }
    uint32_t hit[4];
    HIP_TRY(hipMemcpy(hit, b + 64, 16, hipMemcpyDeviceToHost));
    scratch.release();
    memset(out, 0, sizeof *out);
    if ((int32_t)hit[1] >= 0) {
        out->hit = 1; out->item_index = hit[1]; out->object_id = s->h_items[hit[1]].id;
        memcpy(&out->distance, &hit[0], 4);
    }
    return RR_OK;
} RR_GUARD_END("rr_pick")

// ---------------------------------------------------------------------------
// ray queries: Raytracing::trace for caller-supplied rays (the closest-hit kernel of the deeper levels on a queue that
// the host fills), rr_pick generalised
// ---------------------------------------------------------------------------
extern "C" int rr_trace_rays(rr_scene* s, const float* origins, const float* directions, uint32_t n, uint32_t depth, rr_ray_hit* out) try {
    if (!s || (n && (!origins || !directions || !out))) return fail(RR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (depth == 0 || depth > 255u) return fail(RR_ERR_INVALID_ARGUMENT, "depth %u (1 .. 255)", depth);
    if (n == 0) return RR_OK;
    if (n > 0x7fffff00u) return fail(RR_ERR_UNSUPPORTED, "%u rays in one call", n);
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    RR_FAULT_POINT("trace_rays.host");
    std::vector<float4> r0(n), r1(n);
    std::vector<uint2> r2(n);
    {
        double need[3] = {0.0, 0.0, 0.0};
        for (uint32_t i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) {
                const double a = std::fabs((double)origins[3 * (size_t)i + c]) * 1.001;
                if (std::isfinite(a)) need[c] = std::max(need[c], a);
            }
        int rc = ensure_tlas_reach(s, need);
        if (rc != RR_OK) return rc;
    }
    for (uint32_t i = 0; i < n; i++) {
        r0[i] = make_float4(origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2], 1.0f);
        r1[i] = make_float4(directions[3 * (size_t)i], directions[3 * (size_t)i + 1], directions[3 * (size_t)i + 2], 0.0f);
        r2[i] = make_uint2(depth << 16, 1u);
    }
    DevBuf b0, b1, b2, bh, bc;
    HIP_TRY(b0.reserve((size_t)n * 16)); HIP_TRY(b1.reserve((size_t)n * 16)); HIP_TRY(b2.reserve((size_t)n * 8)); HIP_TRY(bh.reserve((size_t)n * 16));
    HIP_TRY(bc.reserve(256 + sizeof(DShadeConst))); // [0] the level's size, [4] the fetch head, [128] work counters, [256] scene view + (empty) frame constants
    HIP_TRY(hipMemcpy(b0.p, r0.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b1.p, r1.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b2.p, r2.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    uint32_t words[64] = {n, 0u};
    HIP_TRY(hipMemcpy(bc.p, words, sizeof words, hipMemcpyHostToDevice));
    DRayQueue q{b0.as<float4>(), b1.as<float4>(), b2.as<uint2>(), bh.as<uint4>()};
    {
        DShadeConst hc;
        memset(&hc, 0, sizeof hc);
        hc.sc = s->view;
        HIP_TRY(hipMemcpy(bc.as<char>() + 256, &hc, sizeof hc, hipMemcpyHostToDevice));
    }
    DPrimary pr{nullptr, 0ull, 0u, 1u};
    { const int rc = launch_trace_closest(s, false, q, bc.as<uint32_t>(), bc.as<uint32_t>() + 1, n, (const DShadeConst*)(bc.as<char>() + 256), nullptr, pr,
                                          (unsigned long long*)(bc.as<char>() + 128), nullptr);
      if (rc != RR_OK) return rc; }
    std::vector<uint4> hits(n);
    HIP_TRY(hipMemcpy(hits.data(), bh.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        rr_ray_hit& h = out[i];
        memset(&h, 0, sizeof h);
        h.item_index = 0xffffffffu;
        if ((int32_t)hits[i].y >= 0) {
            const DItem& it = s->h_items[hits[i].y];
            h.hit = 1u; h.item_index = hits[i].y; h.object_id = it.id;
            memcpy(&h.distance, &hits[i].x, 4);
            if (!(it.flags & RR_IF_SPHERE)) { // leaf-order slot + side bits -> the reference's face id
                const uint32_t slot = hits[i].z & 0x3fffffffu, back = hits[i].z >> 31;
                h.face_id = s->h_slot_face[it.tri_base + slot] + (back ? it.n_tris : 0u);
            }
        }
    }
    return RR_OK;
} RR_GUARD_END("rr_trace_rays")

// ---------------------------------------------------------------------------
// device arithmetic probe (tests/test_device_math.py): runs rr_math.h functions on the GPU
// ---------------------------------------------------------------------------
extern "C" int rr_math_probe(int op, const float* a, const float* b, const float* c, int n, float* out0, float* out1, float* out2,
                             uint64_t seed, int device) try {
    if (n <= 0 || !a || !out0) return fail(RR_ERR_INVALID_ARGUMENT, "bad arguments");
    if (op == 11) { // the HOST build of the per-triangle shading constants (tri_shading_constants): a, b, c hold n / 3 triangles' vertices, xyz interleaved
        for (int t = 0; t + 2 < n; t += 3) {
            float ng[3], area;
            tri_shading_constants(a + t, b + t, c + t, ng, &area);
            for (int k = 0; k < 3; k++) { out0[t + k] = ng[k]; if (out1) out1[t + k] = area; }
        }
        return RR_OK;
    }
    if (op == 6) { // the HOST build of rr_cos, as make_dmaterial uses it for DMaterial::cos_*: out0[i] = rr_cos(a[i] * pi); needs no device
        for (int i = 0; i < n; i++) out0[i] = rr_cos(a[i] * RR_PI_F);
        return RR_OK;
    }
    HIP_TRY(hipSetDevice(device));
    DevBuf in[3], o[3];
    const float* src[3] = {a, b, c};
    float* dst[3] = {out0, out1, out2};
    for (int k = 0; k < 3; k++) {
        HIP_TRY(in[k].reserve((size_t)n * 4)); HIP_TRY(o[k].reserve((size_t)n * 4));
        if (src[k]) HIP_TRY(hipMemcpy(in[k].p, src[k], (size_t)n * 4, hipMemcpyHostToDevice));
        else HIP_TRY(hipMemset(in[k].p, 0, (size_t)n * 4));
        HIP_TRY(hipMemset(o[k].p, 0, (size_t)n * 4));
    }
    hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, nullptr, op, in[0].as<float>(), in[1].as<float>(), in[2].as<float>(), n,
                       o[0].as<float>(), o[1].as<float>(), o[2].as<float>(), (uint32_t)seed, (uint32_t)(seed >> 32));
    HIP_TRY(hipDeviceSynchronize());
    for (int k = 0; k < 3; k++) {
        if (dst[k]) HIP_TRY(hipMemcpy(dst[k], o[k].p, (size_t)n * 4, hipMemcpyDeviceToHost));
        in[k].release(); o[k].release();
    }
    return RR_OK;
} RR_GUARD_END("rr_math_probe")

#ifdef RR_EXP_UTIL
extern "C" int rr_exp_util(unsigned long long* out64, int reset) {
    if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_util), sizeof(g_util)) != hipSuccess) return RR_ERR_DEVICE;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_util), z, sizeof(z)) != hipSuccess) return RR_ERR_DEVICE; }
    return RR_OK;
}
#endif
