/*
 * rustray_hip.h — C ABI of the MI355X trace-loop replacement for rustray.
 *
 * This is the drop-in boundary for ONE path of the reference: the per-pixel /
 * per-sample trace loop that `RendererManager::start` spawns
 * (reference src/renderer.rs:105-172, worker body :267-315) and that calls
 * `Raytracing::render(x, y) -> PixelData` (reference src/raytracing.rs:275-427).
 * One `rr_render*` call replaces that whole cell-queue + thread-pool frame and
 * fills the four buffers `Run::apply_pixels` fills today
 * (reference src/run.rs:519-541: image RGBA8, normals, depth, objects).
 *
 * The reference has no FFI; its `Scene` is made of Rust trait objects
 * (src/scene.rs:69-83, src/shape/mod.rs:14-46).  The boundary therefore takes
 * a *flat scene*: plain arrays that a small host-side shim fills by walking
 * `Scene` (see INTEGRATION.md for the Rust shim).  Every struct below is POD,
 * little-endian, naturally aligned; all matrices are column-major 4x4 f32 as
 * nalgebra stores them.
 *
 * Ownership: the caller owns every input and output buffer.  The library
 * copies what it needs during rr_scene_create and owns only its handle and
 * device memory.  Nothing here calls back into the host.  No function aborts
 * or throws across the ABI: every entry point returns RR_OK (0) or a negative
 * rr_status, and rr_last_error() returns a thread-local message.
 */
#ifndef RUSTRAY_HIP_H
#define RUSTRAY_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's struct layouts and entry points.  rr_scene_create refuses a flat scene that names another
 * version, and rr_abi_version() reports what the loaded library was built with, so a caller compiled against an older
 * header (round 1: a shorter rr_frame_stats, rr_scene_set_profiling) fails at the first call instead of being written
 * past its structs.  2: rr_frame_stats grew (level-1 timing, binning, multi-GPU exchange), rr_tuning, rr_abi_version.
 * 3: rr_frame_stats carries the level-1 share of the shade and shadow kernels too (one roofline per kernel build in bench.py). */
#define RR_ABI_VERSION 3u

typedef enum rr_status {
    RR_OK = 0,
    RR_ERR_INVALID_ARGUMENT = -1,  /* NULL pointer, bad index, bad size */
    RR_ERR_UNSUPPORTED = -2,       /* e.g. max_recursion above RR_MAX_RECURSION */
    RR_ERR_NO_DEVICE = -3,         /* no HIP device / HIP runtime failure at init */
    RR_ERR_DEVICE = -4,            /* a HIP call failed during the frame */
    RR_ERR_OUT_OF_MEMORY = -5,
    RR_ERR_CANCELLED = -6          /* *cancel became non-zero during the frame */
} rr_status;

/* Largest RaytracingConfig::max_recursion the device path accepts (reference default: 6, src/raytracing.rs:124).  A path node at
 * recursion depth d carries d in five bits of its shadow-ray records and its index (root 1, children 2k and 2k + 1) in 32: depth
 * max_recursion + 1 <= 31. */
#define RR_MAX_RECURSION 30u

/* Largest RaytracingConfig::samples accepted.  The reference computes the sub-sample cell size as
 * `(samples + 2).next_power_of_two() / 2` in u16 arithmetic (src/raytracing.rs:297), which overflows from 32767 samples on:
 * RR_MAX_SAMPLES_WITH_TABLE = 32766 is the reference's own limit, accepted whenever the caller passes its sub-sample table
 * (`sample_xy`, the recommended way: INTEGRATION.md).  The BUILT-IN table (sample_xy = NULL, rr_sample_table) is limited to
 * RR_MAX_SAMPLES = 16382, the last count of the cell size below that (cell_size 8192: 8192^2 = 67 M cells, 268 MB of host memory
 * while they are shuffled; the next cell size needs 1 GB). */
#define RR_MAX_SAMPLES_WITH_TABLE 32766u
#define RR_MAX_SAMPLES 16382u

/* Largest number of scene items (the reference has none: `items: Vec<..>`, src/scene.rs:69-83).  The top level of the acceleration
 * structure holds one item per leaf and shares a fixed traversal stack (39 entries per ray, in LDS) with the per-mesh trees: up to
 * 4 096 items it takes 12 levels of it, beyond that ceil(log2 n_items), which leaves 36 - ceil(log2 n_items) levels -- at least 16:
 * 524 288 triangles in any ONE mesh at worst, 134 M with up to 4 096 items -- to a mesh's own tree.  rr_scene_create answers
 * RR_ERR_UNSUPPORTED beyond either bound.  Scenes of 17 .. 512 items additionally get the packet form of the top level (coherent
 * packets test the items' boxes with one item per lane instead of walking the tree per ray); outside that range every ray walks
 * the tree -- same frame, bit for bit, at more node steps per ray.
 * Memory per INSTANCE of a mesh: a 208-B item record plus 32 B per triangle of the mesh (the world normals of its flat-shaded hits,
 * both signs, evaluated once per instanced triangle instead of per hit); the mesh itself -- vertices, attributes, tree: about 230 B
 * per triangle -- is shared by its instances.  The sum over all items of their meshes' triangle counts is limited to 2^31 (64 GB of
 * such normals): RR_ERR_UNSUPPORTED beyond. */
#define RR_MAX_ITEMS 1048576u

/* TextureType order of reference src/shape/mod.rs:633-643. */
enum {
    RR_TEX_BASE = 0,
    RR_TEX_AMBIENT_EMISSIVE = 1,
    RR_TEX_SPECULAR = 2,
    RR_TEX_NORMAL = 3,
    RR_TEX_ALPHA = 4,
    RR_TEX_ROUGHNESS = 5,
    RR_TEX_AMBIENT_OCCLUSION = 6,
    RR_TEX_REFLECTIVITY = 7,
    RR_TEX_COUNT = 8
};

/* One decoded image (image::DynamicImage after `.to_rgba()`,
 * reference src/shape/mod.rs:531, :586-589).  Row-major, top row first. */
typedef struct rr_texture {
    uint32_t width;
    uint32_t height;
    const uint8_t* rgba8; /* width*height*4 bytes */
} rr_texture;

/* All fields of `Material` that the trace loop reads
 * (reference src/shape/mod.rs:95-134; defaults :138-180). */
typedef struct rr_material {
    float ambient_color[3];
    float base_color[3];
    float specular_color[3];
    float alpha;
    float shininess;
    float reflectivity;
    float refraction_index;
    float normal_map_strength;
    float shadow_softness;
    float roughness;
    int32_t texture[RR_TEX_COUNT]; /* index into rr_flat_scene.textures, or -1 */
    uint8_t texture_filtering_nearest;
    uint8_t cast_shadow;
    uint8_t receive_shadow;
    uint8_t monte_carlo;
    uint8_t smooth_shading;
    uint8_t reflection_only;
    uint8_t backface_cullig; /* sic: the reference's field name */
    uint8_t _pad;
} rr_material;

/* Triangle mesh data of `Mesh` (reference src/shape/mesh.rs:10-21).  Meshes
 * are shared: several items may name the same mesh (instancing). */
typedef struct rr_mesh {
    const float* positions;         /* n_vertices * 3 */
    const uint32_t* indices;        /* n_triangles * 3 */
    const float* uvs;               /* n_uvs * 2, may be NULL */
    const uint32_t* uv_indices;     /* n_uv_faces * 3, may be NULL */
    const float* normals;           /* n_normals * 3, may be NULL */
    const uint32_t* normal_indices; /* n_normal_faces * 3, may be NULL */
    uint32_t n_vertices;
    uint32_t n_triangles;
    uint32_t n_uvs;
    uint32_t n_uv_faces;
    uint32_t n_normals;
    uint32_t n_normal_faces;
} rr_mesh;

enum { RR_ITEM_SPHERE = 0, RR_ITEM_MESH = 1 };

/* One scene item: `ShapeBasics` (reference src/shape/mod.rs:661-680) plus
 * the shape payload.  Item ORDER is semantically significant
 * (reference src/raytracing.rs:440-487) and must be Scene.items order. */
typedef struct rr_item {
    uint32_t kind;          /* RR_ITEM_SPHERE | RR_ITEM_MESH */
    uint32_t id;            /* ShapeBasics::id, reported as object_id */
    int32_t material;       /* `get_material()`: full material, textures included */
    int32_t material_cache; /* `get_material_cache_without_textures()`
                               (reference src/shape/mod.rs:769-772): must have every
                               texture slot = -1 */
    int32_t mesh;           /* index into meshes (kind == mesh), else -1 */
    float radius;           /* Ball radius (kind == sphere) */
    float trans[16];        /* ShapeBasics::trans, column-major */
    float trans_inv[16];    /* ShapeBasics::tran_inverse, column-major */
    float bbox_min[3];      /* ShapeBasics::b_box in LOCAL space */
    float bbox_max[3];
    uint8_t visible;
    uint8_t flip_normals;
    uint8_t _pad[2];
} rr_item;

enum { RR_LIGHT_DIRECTIONAL = 0, RR_LIGHT_POINT = 1, RR_LIGHT_SPOT = 2 };

/* `Light` (reference src/scene.rs:40-51). */
typedef struct rr_light {
    float pos[3];
    float dir[3];
    float color[3];
    float intensity;
    float max_angle; /* radians */
    uint32_t light_type;
    uint8_t enabled;
    uint8_t _pad[3];
} rr_light;

typedef struct rr_flat_scene {
    uint32_t abi_version; /* RR_ABI_VERSION */
    uint32_t n_items;
    uint32_t n_meshes;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_lights;
    const rr_item* items;
    const rr_mesh* meshes;
    const rr_material* materials;
    const rr_texture* textures;
    const rr_light* lights;
} rr_flat_scene;

/* What the trace loop reads of `Camera` (reference src/camera.rs:19-40,
 * used at src/raytracing.rs:282-283, :340, :349, :355-356, :369-393). */
typedef struct rr_camera {
    uint32_t width;
    uint32_t height;
    float projection_inverse[16];
    float view_inverse[16];
} rr_camera;

/* `RaytracingConfig` by value (reference src/raytracing.rs:92-106) plus the
 * seed of the counter-based RNG that replaces the reference's un-seeded
 * `rand::thread_rng()` in `jitter` (src/raytracing.rs:616-618).  Pass the
 * EFFECTIVE config: JSON `config` blocks override the CLI in the reference
 * (src/scene.rs:180-198). */
typedef struct rr_config {
    uint64_t seed;
    float focal_length;
    float aperture_size;
    float fog_density;
    float fog_color[3];
    uint16_t samples;
    uint16_t max_recursion;
    uint8_t monte_carlo;
    uint8_t gamma_correction;
    uint8_t _pad[2];
} rr_config;

/* Output of one frame = what `Run::apply_pixels` stores per PixelData
 * (reference src/run.rs:519-541, PixelData src/raytracing.rs:57-70).
 * rgba8 is required; the aux pointers may be NULL.  All are row-major with
 * `width` pixels per row, y = 0 at the top. */
typedef struct rr_frame {
    uint8_t* rgba8;      /* w*h*4, alpha forced to 255 (src/run.rs:527) */
    float* normal;       /* w*h*3 */
    float* depth;        /* w*h */
    uint32_t* object_id; /* w*h */
} rr_frame;

/* A subset of the frame for one rank of a multi-GPU render.  The frame is cut
 * into tiles of tile_w x tile_h pixels, numbered row-major; this rank owns the
 * tiles with (tile_index % n_ranks == rank).  The rank's pixels are written
 * COMPACTLY, in tile order then row-major inside the tile (clipped at the
 * frame border), into buffers of rr_region_pixel_count() pixels.
 * n_ranks = 1, rank = 0 selects the whole frame (still in tile order). */
typedef struct rr_region {
    uint32_t tile_w;
    uint32_t tile_h;
    uint32_t n_ranks;
    uint32_t rank;
} rr_region;

/* Result of rr_pick (reference Raytracing::pick, src/raytracing.rs:237-273). */
typedef struct rr_pick_result {
    uint32_t hit;       /* 0 = None */
    uint32_t object_id; /* ShapeBasics::id */
    uint32_t item_index;
    float distance;
} rr_pick_result;

/* Per-frame work counters: one "ray" = one Raytracing::trace
 * call (reference src/raytracing.rs:429). */
typedef struct rr_frame_stats {
    uint64_t primary_rays;
    uint64_t secondary_rays; /* reflection + refraction */
    uint64_t shadow_rays;
    uint64_t shaded_hits;
    double ms_total;        /* device time of the whole frame */
    double ms_trace_closest;
    double ms_trace_shadow;
    double ms_shade;
    uint64_t launches_trace_closest;
    uint64_t launches_trace_shadow;
    uint64_t launches_shade;
    uint64_t batches;       /* device batches of primary samples the frame was cut into */
    uint64_t sliced_levels; /* depth levels whose children did not fit behind them in the ray arena at once */
    uint64_t binned_rays;   /* secondary rays that were re-ordered by (origin cell, direction octant) before being traced */
    double ms_binning;      /* device time of that re-ordering (kernel_timing) */
    double ms_trace_closest_level1;          /* the part of ms_trace_closest spent on depth level 1 (the primary rays) */
    uint64_t launches_trace_closest_level1;
    /* rr_render_multi only (on scenes[0]; zero after any other frame): how the per-device buffers reached scenes[0]'s device */
    uint32_t multi_devices;       /* handles that took part */
    uint32_t multi_peer_links;    /* handles whose buffers went device-to-device (peer access enabled both ways, or the same device) */
    uint32_t multi_staged_links;  /* handles whose buffers were staged through pinned host memory (no peer access between the devices) */
    uint32_t _pad;
    double ms_multi_exchange;     /* host wall time from the last device finishing its tiles to the frame being in `out` */
    /* the share of ms_shade / ms_trace_shadow spent in the level-1 BUILDS of those kernels (k_shade<true>; k_trace_shadow<true>: level 1 of a scene
     * whose shadow rays have fixed slots -- 17 .. 512 items, up to 32 enabled lights -- else 0: level 1 then runs the deeper levels' build), as
     * ms_trace_closest_level1 (ABI 3) */
    double ms_shade_level1;
    uint64_t launches_shade_level1;
    double ms_trace_shadow_level1;
    uint64_t launches_trace_shadow_level1;
} rr_frame_stats;

/* Execution knobs of the device path.  None of them changes a single output bit (fixed-point accumulation makes
 * the frame independent of batching, chunking and grouping); they exist for memory-constrained hosts, for tests
 * that force the slicing paths, and for profiling.  All zero = automatic.  The library reads NO environment
 * variables. */
typedef struct rr_tuning {
    uint32_t struct_size;        /* sizeof(rr_tuning) */
    uint32_t sample_group;       /* primary samples of one pixel per 64-ray packet: 0 = largest that divides `samples`, else 1, 2, 4 ... 64 */
    uint64_t queue_budget_bytes; /* ray-arena memory: 0 = a quarter of the free HBM, at most 64 GB */
    uint64_t shade_chunk_rays;   /* rays shaded per launch: 0 = 64 Mi (minimum 65536) */
    uint32_t kernel_timing;      /* non-zero: per-launch HIP events fill the ms_* fields of rr_frame_stats */
    uint32_t multi_force_staged; /* rr_render_multi, read from scenes[0]: non-zero stages every device's buffers through pinned host memory even
                                    where peer access exists (the path taken between devices WITHOUT peer access; lets one GPU exercise it) */
    uint64_t bin_min_rays;       /* deeper depth levels of at least this many rays are re-ordered by (origin cell, direction
                                    octant) before they are traced: 0 = never (measured: spawn order is already coherent) */
} rr_tuning;

typedef struct rr_scene rr_scene; /* opaque */

/* RR_ABI_VERSION of the loaded library (never fails). */
uint32_t rr_abi_version(void);

/* Number of HIP devices visible; 0 if none (never fails). */
int rr_device_count(void);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* rr_last_error(void);

/* Validate and upload a flat scene to `device`; build the acceleration
 * structures (replaces Scene::update's BVH build, reference
 * src/scene.rs:1674-1688, and parry's per-TriMesh Qbvh, src/shape/mesh.rs:171). */
int rr_scene_create(const rr_flat_scene* scene, int device, rr_scene** out);
void rr_scene_destroy(rr_scene* scene);

/* Replace item transforms in place (animation / GUI edits between frames:
 * reference ShapeBasics::apply_mat, src/shape/mod.rs:748-753; Scene::apply_frame
 * src/scene.rs:1695-1713).  trans / trans_inv hold n_items * 16 floats. */
int rr_scene_update_transforms(rr_scene* scene, const float* trans, const float* trans_inv);

/* Replace every material in place (GUI edits between frames: reference Material::apply_diff, src/shape/mod.rs:182-242,
 * driven from src/run.rs:1132-1133).  `materials` holds the same n_materials records, in the same order, as the
 * flat scene the handle was created from (full materials and material caches alike); texture slots may name any
 * texture uploaded at creation.  Meshes, acceleration structures and texture images are not touched. */
int rr_scene_update_materials(rr_scene* scene, const rr_material* materials, uint32_t n_materials);

/* Compatibility switches: behaviours of EARLIER reference binaries that the source at HEAD no longer has.  Default 0 = HEAD.
 * RR_COMPAT_OCCLUDER_ALPHA_SHADOWS: a shadow is attenuated by the OCCLUDER's material.alpha, where HEAD takes the
 * receiver's (`shadow_source_alpha = material.alpha`, src/raytracing.rs:898).  That is what the binary behind the 2022-05
 * README renderings did (tests/test_ref_shots.py: with it the product matches those renderings over the whole frame);
 * it exists so that the shipped kernels can be checked against the only outputs the reference holds. */
#define RR_COMPAT_OCCLUDER_ALPHA_SHADOWS 1u
int rr_scene_set_compat(rr_scene* scene, uint32_t flags);

/* Execution knobs (see rr_tuning). */
int rr_scene_set_tuning(rr_scene* scene, const rr_tuning* tuning);
int rr_scene_get_tuning(const rr_scene* scene, rr_tuning* tuning);

/* The reference's per-pixel sub-sample table (src/raytracing.rs:290-313):
 * cell_size^2 cells shuffled with StdRng::seed_from_u64(0), truncated to
 * `samples`.  Writes samples*2 uint16 (x_i, y_i) and *cell_size. */
int rr_sample_table(uint16_t samples, uint16_t* xy_out, uint32_t* cell_size_out);

/* Render one whole frame into HOST buffers (synchronous).
 * sample_xy: samples*2 uint16 from the host's own rand, or NULL for the
 * built-in table.  cancel: optional flag polled between device launches. */
int rr_render(rr_scene* scene, const rr_camera* camera, const rr_config* config,
              const uint16_t* sample_xy, const rr_frame* out, const volatile int* cancel);

/* The same frame on SEVERAL GPUs from one host process, as the reference host is one process (src/renderer.rs:105-172).
 * scenes[i] is a handle created with rr_scene_create(scene, device_i, ...) from the SAME flat scene; handle i renders
 * the 32x8-pixel tiles with (tile_index % n_scenes == i) on its device (one host thread per device inside the call),
 * the compact per-device buffers are copied peer-to-peer into scenes[0]'s device, de-interleaved there and copied to
 * `out` (host buffers, as rr_render).  The frame is bit-identical to rr_render's for any n_scenes.  Two handles may sit
 * on the same device (a rehearsal on one GPU).  Peer access between scenes[0]'s device and every other device is
 * checked (hipDeviceCanAccessPeer) and enabled both ways on first use; a pair without it is staged through pinned host
 * memory instead, and rr_scene_last_stats(scenes[0]) says which way each handle's buffers went.  Every device works on
 * a non-blocking stream of its own.  Handles are locked in address order: concurrent calls that share handles, in any
 * order, serialise instead of deadlocking. */
int rr_render_multi(rr_scene* const* scenes, uint32_t n_scenes, const rr_camera* camera, const rr_config* config,
                    const uint16_t* sample_xy, const rr_frame* out, const volatile int* cancel);

/* Diagnostic: the order in which rr_render_multi locks `scenes` (address order), as indices into the caller's array.
 * Looks at the pointer values only, never through them (tests/test_abi.py checks the order without a GPU). */
int rr_multi_lock_order(rr_scene* const* scenes, uint32_t n_scenes, uint32_t* order_out);

/* Progressive form of rr_render.  Stands in for the progressive fill the reference shows while a frame renders:
 * Run::apply_pixels drains the PixelData channel every GUI tick (src/run.rs:506-545) and RendererManager::stop
 * (src/renderer.rs:174-198) ends the frame early.  The frame is rendered in at least `min_passes` device batches of
 * whole sample slices (every pixel, a subset of the samples); after each batch but the last, the host buffers of
 * `out` hold the frame resolved over the samples finished so far (colour, normal and depth are means; object_id is
 * only final after the last batch) and `on_pass(user, samples_done, samples_total)` is called on the calling
 * thread.  A non-zero return stops the frame: the call returns RR_ERR_CANCELLED and `out` keeps the last preview.
 * The finished frame is bit-identical to rr_render's. */
typedef int (*rr_pass_fn)(void* user, uint64_t primary_samples_done, uint64_t primary_samples_total);
int rr_render_progressive(rr_scene* scene, const rr_camera* camera, const rr_config* config,
                          const uint16_t* sample_xy, const rr_frame* out, uint32_t min_passes,
                          rr_pass_fn on_pass, void* user, const volatile int* cancel);

/* The frame filled in TILE BY TILE instead: every pixel is final when it appears, as in the reference's GUI (shuffled 2x2-pixel cells, each
 * rendered with all of its samples: src/renderer.rs:125-172).  Pass k of n_passes (0 = 16; at most the number of tiles) renders the 32x8-pixel
 * tiles with (tile_index % n_passes == k), an interleaved subset of the frame, and after each pass but the last the host buffers of `out` hold the
 * frame so far (pixels not rendered yet are zero) and `on_pass(user, samples_done, samples_total)` is called on the calling thread; a non-zero
 * return stops the frame (RR_ERR_CANCELLED, `out` keeps what was finished).  The finished frame is bit-identical to rr_render's;
 * rr_scene_last_stats reports the sums over the passes. */
int rr_render_progressive_tiles(rr_scene* scene, const rr_camera* camera, const rr_config* config,
                                const uint16_t* sample_xy, const rr_frame* out, uint32_t n_passes,
                                rr_pass_fn on_pass, void* user, const volatile int* cancel);

/* Number of pixels `region` owns in a width x height frame. */
uint64_t rr_region_pixel_count(uint32_t width, uint32_t height, const rr_region* region);

/* Render `region` of the frame into DEVICE buffers (HIP device pointers on the
 * scene's device) holding rr_region_pixel_count() pixels, compact, in region
 * order.  Work is enqueued on `hip_stream` (a hipStream_t, NULL = the default
 * stream); the call returns once everything is enqueued or, when the frame
 * needs more than one batch of samples, after the last batch has been
 * enqueued (earlier batches are awaited).  The caller synchronises the stream. */
int rr_render_region_device(rr_scene* scene, const rr_camera* camera, const rr_config* config,
                            const uint16_t* sample_xy, const rr_region* region,
                            const rr_frame* out_device, void* hip_stream,
                            const volatile int* cancel);

/* Scatter a compact region buffer back to full-frame layout, on device.
 * Used by rank 0 after the gather; src holds the concatenated per-rank
 * compact buffers in rank order. elem_bytes = bytes per pixel (4, 12, 4, 4). */
int rr_deinterleave_device(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                           uint32_t n_ranks, uint32_t elem_bytes, const void* src_device,
                           void* dst_device, int device, void* hip_stream);

/* The same for the gathered PACKS of a multi-rank frame, all buffers in one launch and without a concatenation pass: rank r's pack
 * (a byte buffer) starts at packs + r * pack_stride and holds, for output buffer k (0 rgba8, 1 normal, 2 depth, 3 object id), that
 * rank's compact pixels at section_offset[k] (elem_bytes[k] bytes per pixel; 0 = the buffer is absent, dst[k] ignored).  dst[k]: the
 * frame-order buffer.  This is what rank 0 runs on the target of its one gather per frame (rustray_amd/renderer.py: TiledFrame). */
int rr_deinterleave_packed_device(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks,
                                  const void* packs_device, uint64_t pack_stride, const uint64_t* section_offset,
                                  const uint32_t* elem_bytes, void* const* dst_device, int device, void* hip_stream);

/* Single-ray query at the pixel centre (reference src/raytracing.rs:237-273). */
int rr_pick(rr_scene* scene, const rr_camera* camera, int x, int y, rr_pick_result* out);

/* Closest-hit queries for a batch of caller-supplied rays: Raytracing::trace(ray, false, false, depth)
 * (reference src/raytracing.rs:429-490) for each of them, the generalisation of rr_pick (which is one such query for
 * a pixel-centre ray).  origins / directions: n * 3 floats (host); directions are used as given (trace does not
 * normalise).  `depth` is the recursion depth the candidate filter sees: reflection-only items are candidates for
 * depth > 1 only (:454).  out[i].item_index = 0xffffffff when nothing is hit. */
typedef struct rr_ray_hit {
    uint32_t hit;        /* 0 = None */
    uint32_t item_index;
    uint32_t object_id;  /* ShapeBasics::id of the item */
    uint32_t face_id;    /* as Shape::intersect reports it: triangle index, + n_triangles for back faces; 0 for spheres */
    float distance;      /* toi */
} rr_ray_hit;
int rr_trace_rays(rr_scene* scene, const float* origins, const float* directions, uint32_t n, uint32_t depth, rr_ray_hit* out);

/* Post-processing of a finished frame (reference run_post_processing, src/post_processing.rs:123-181, called from
 * Run::post_processing, src/run.rs:588-600): outline on object-id edges (:98-121), then cavity = curvature of the
 * normal buffer (:77-96), clamp, truncate to u8.  Consumes exactly the buffers rr_render produces.
 * rr_post_process: host buffers; rr_post_process_device: device buffers on `device`, enqueued on `hip_stream`.
 * rgba_in and rgba_out may not alias (the reference writes a new image). */
int rr_post_process(uint32_t width, uint32_t height, int cavity, int outline, const uint8_t* rgba_in,
                    const float* normal, const uint32_t* object_id, uint8_t* rgba_out, int device);
int rr_post_process_device(uint32_t width, uint32_t height, int cavity, int outline, const uint8_t* rgba_in,
                           const float* normal, const uint32_t* object_id, uint8_t* rgba_out, int device,
                           void* hip_stream);

/* Counters and device timings of the most recent frame on this scene. */
int rr_scene_last_stats(const rr_scene* scene, rr_frame_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* RUSTRAY_HIP_H */
