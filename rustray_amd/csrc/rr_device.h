// rr_device.h — HBM-resident data layout of a scene and of the ray queues.
// Shared by the host uploader (rr_api.cpp / rr_bvh.cpp) and the kernels.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// ---- acceleration structures -------------------------------------------------------------
// BVH2 node, 64 B, one cache line per visit.  Both children's boxes live in the
// parent, so a traversal step is one 4 x dwordx4 fetch.
//   n0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//   n1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   n2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//   n3 = (child0, child1, 0, 0)   as int bits
// child >= 0: inner node index (relative to the structure's node base)
// child <  0: leaf, ~child = first | (count-1) << 28  (first relative to the prim base)
struct DNode { float4 n0, n1, n2, n3; };

// BVH4 node (top level and per-mesh trees), 128 B = one cache line: four child boxes in SoA form and four child codes.
//   q0 = lo.x[0..3]  q1 = hi.x[0..3]  q2 = lo.y  q3 = hi.y  q4 = lo.z  q5 = hi.z  q6 = child codes (int bits)  q7 unused
// An unused slot is a point at +infinity (its entry distance is +inf or its exit -inf, never a hit).
// (A 64-B form with the child boxes quantised to 8 bits on a node-local grid was built and measured in round 2: 30 %
// fewer L1 accesses, 19 % more vector instructions for the decode, closest-hit time equal and shadow time +8 %: dropped.)
struct DNode4 { float4 q[8]; };

#define RR_LEAF_FIRST(code) ((uint32_t)(code) & 0x0fffffffu)
#define RR_LEAF_COUNT(code) ((((uint32_t)(code)) >> 28) + 1u)
#define RR_MAX_LEAF_TRIS 8
// Depth limits enforced by the host builder, so the fixed LDS stack can never overflow: per level one
// sentinel entry plus at most one pending sibling per inner node on the path, plus one scratch slot above the
// top (the node step writes the far child before it knows whether it is needed).
#ifndef RR_BLAS_MAX_DEPTH
#define RR_BLAS_MAX_DEPTH 24
#endif
#ifndef RR_TLAS_MAX_DEPTH
#define RR_TLAS_MAX_DEPTH 12
#endif
#ifndef RR_STACK_DEPTH
#define RR_STACK_DEPTH (RR_BLAS_MAX_DEPTH + RR_TLAS_MAX_DEPTH + 3)
#endif

// Triangle as the shading of a hit reads it, 64 B (4 x dwordx4), in BVH leaf order:
//   v0 = (a.xyz, bits(original face index)), v1 = (b.xyz, area), v2 = (c.xyz, 0), v3 = (flat normal.xyz, 0)
// area = |cross(a - b, a - c)| (Mesh::get_normal / get_uv divide their weights by it, src/shape/mesh.rs:127-143) and the flat normal
// normalize(cross(b - a, c - a)) (parry's triangle normal, mesh.rs:76-98) are the same IEEE values for every hit of the triangle:
// the host evaluates them once, with the sequence k_shade used per hit (rr_api.hip tri_shading_constants; a flat, untextured hit
// then reads v3 alone).
struct DTri { float4 v0, v1, v2, v3; };

// The same triangle as the walks test it, 48 B = 3 x dwordx4, same order: the edge vectors of parry's ray/triangle
// test are the same IEEE values for every ray, so the host computes them once (no contraction); the plane normal
// n = cross(ab, ac) costs nine instructions per test, a fourth 16-B load per lane costs more (L1 return path).
//   t0 = (a.xyz, bits(original face index)), t1 = (ab.xyz, ac.x), t2 = (ac.y, ac.z, 0, 0)     ab = b - a, ac = c - a
struct DTriX { float4 t0, t1, t2; };

// Per-triangle shading attributes, 64 B, same order as DTri (fetched once per shaded hit):
//   s0 = (n0.xyz, uv0.x) s1 = (n1.xyz, uv0.y) s2 = (n2.xyz, uv1.x) s3 = (uv1.y, uv2.x, uv2.y, bits(flags))
// flags bit0: the reference's get_uv finds uv indices for this face (src/shape/mesh.rs:116)
struct DTriAttr { float4 s0, s1, s2, s3; };

// ---- items ------------------------------------------------------------------------------
enum : uint32_t {
    RR_IF_VISIBLE = 1u << 0,        // ShapeBasics::visible
    RR_IF_FLIP_NORMALS = 1u << 1,   // ShapeBasics::flip_normals
    RR_IF_CACHE_ALPHA_POS = 1u << 2,   // material_cache.alpha > 0
    RR_IF_CACHE_CAST_SHADOW = 1u << 3, // material_cache.cast_shadow
    RR_IF_CACHE_REFL_ONLY = 1u << 4,   // material_cache.reflection_only
    RR_IF_SOLID_BASE = 1u << 5,     // !(cache.alpha < 1 || cache has alpha tex) && cache.backface_cullig
    RR_IF_SMOOTH = 1u << 6,         // cache.smooth_shading && mesh has normals and normal indices
    RR_IF_SPHERE = 1u << 7,
    RR_IF_OCCLUDER_ALPHA_TEX = 1u << 8, // full material has an alpha texture (shadow attenuation lookup)
    RR_IF_HAS_UV_FACES = 1u << 9,
    RR_IF_UV_MAY_BE_NAN = 1u << 10, // get_uv of this item can return a non-finite value for a finite point: every sphere (acos beyond 1 away
                                    // from its surface), a mesh with a zero-area triangle (area weights 0/0).  k_shade, want_shadow.
};

// 208 B instance record.  Rows of the column-major matrices, so that
// row4(inv0, x, y, z, w) is exactly nalgebra's (M * v).x.
struct DItem {
    float4 inv0, inv1, inv2, inv3; // trans_inv rows (inv3 = w row)
    float4 tr0, tr1, tr2;          // trans rows 0..2 (normals: trans * (n, 0))
    float bmin[3]; float radius;
    float bmax[3]; uint32_t flags;
    uint32_t id;         // ShapeBasics::id
    int32_t material;    // index into DMaterial[]
    uint32_t wn_base;    // first entry of this item's flat world normals (DSceneView::flat_normals: two per triangle)
    uint32_t _r1;
    uint32_t tri_base;   // first DTri / DTriX / DTriAttr of the mesh
    uint32_t n_tris;
    uint32_t node_base4; // first DNode4 of the mesh's tree
    int32_t root4;       // node index relative to node_base4, a leaf code, or RR_SENTINEL (no triangles)
};

// 240 B material record (seven 16-B groups + eight 16-B texture descriptors)
struct DMaterial {
    float ambient[3]; float alpha;
    float base[3]; float shininess;
    float specular[3]; float reflectivity;
    float refraction_index, normal_map_strength, shadow_softness, roughness;
    int32_t tex[8];
    uint32_t flags; // bit0 nearest filtering, bit1 receive_shadow, bit2 monte_carlo, bit3 any texture, bits 8..15 slot k holds a texture
    // jitter()'s cone bound z_lo = cos(spread * pi) for the two spreads that are material constants, evaluated once on the host with the
    // kernels' own rr_cos (src/raytracing.rs:590-596 computes it per call; a roughness MAP gives a per-hit spread and is evaluated per hit)
    float cos_shadow_softness, cos_roughness;
    uint32_t _pad;
    // the descriptors of the eight texture slots, copied from DSceneView::textures: a texel fetch is then material -> texel instead of
    // material -> texture index -> descriptor -> texel (two dependent round trips fewer per fetch in k_shade's chain)
    struct { uint64_t offset; uint32_t width, height; } texd[8];
};
enum : uint32_t { RR_MF_NEAREST = 1u, RR_MF_RECEIVE_SHADOW = 2u, RR_MF_MONTE_CARLO = 4u, RR_MF_ANY_TEX = 8u, RR_MF_TEX_SLOT0 = 256u };

struct DTexture { uint64_t offset; uint32_t width, height; }; // offset in texels into the RGBA8 pool

struct DLight {
    float pos[3]; float intensity;
    float dir[3]; float max_angle;
    float color[3]; uint32_t type; // RR_LIGHT_*; disabled lights are kept (index = RNG stream) with type | 0x80
};

struct DSceneView {
    const DItem* items;
    const DNode4* nodes4;   // per-mesh trees; DItem::node_base4 / root4 index it
    const DTri* tris;       // vertices, for shading
    const DTriX* trix;      // the walks' form
    const DTriAttr* attrs;
    // The world normal of a flat-shaded hit depends on the item's transform and the triangle only: normalize(trans * (+-flat normal, 0))
    // (Shape::intersect, src/shape/mesh.rs:76-98).  k_world_normals evaluates it once per instanced triangle and sign -- with the very
    // device function k_shade used per hit (to_world_normal) -- at scene creation and after a transform update:
    // flat_normals[item.wn_base + 2 * slot + (negated ? 1 : 0)].xyz.  32 B per instanced triangle.
    const float4* flat_normals;
    const uint32_t* face_slot; // per mesh triangle: original face index -> leaf-order slot
    const DMaterial* materials;
    const DTexture* textures;
    const uint32_t* texels; // RGBA8 pool
    const DLight* lights;
    uint32_t n_items;
    uint32_t n_lights;
    uint32_t n_enabled_lights; // any number; level 1 uses fixed shadow slots for up to RR_FIXED_SLOT_LIGHTS of them
    const DNode4* tnodes4;  // the top level over item world boxes, same form (one item per leaf)
    int32_t tlas_root4;      // node index, a leaf code (one item), or RR_SENTINEL (empty scene)
    const DNode4* tnodes4c;  // the top level of the per-ray CLOSEST-HIT walks: over the items' surface boxes where those are tighter (rr_api.hip build_tlas),
    int32_t tlas_root4c;     // else the same tree as tnodes4 / tlas_root4 (which shadow queries always take)
    const float4* item_boxes; // padded world boxes per item for the packet form of the top level: [2 i] = lo, [2 i + 1] = hi of the item's corner box (the boxes of
                              // the tree; trace_shadow_packet), [2 (n_items + i)], [.. + 1] of its surface box (trace_closest_packet); rr_api.hip build_tlas
    uint32_t any_alpha_occluder; // some item's material has an alpha map: the shadow attenuation of a receiver whose uv may be NaN can be NaN (k_shade, want_shadow)
    uint32_t general_w;      // some trans_inv has a w row other than (0,0,0,1)
    uint32_t compat;         // RR_COMPAT_* (rr_scene_set_compat): behaviours of earlier reference binaries; 0 = the source at HEAD.
                             // RR_VIEW_NAN_BALLS (library-internal, build_tlas): some ball's arithmetic can overflow into a NaN toi
};

// ---- per-frame constants ----------------------------------------------------------------
struct DFrame {
    float proj_inv[16];
    float view_inv[16];
    uint32_t width, height;
    uint32_t samples, cell_size;
    uint32_t max_recursion;
    uint32_t monte_carlo, gamma, dof;
    float focal_length, aperture_size, fog_density;
    float fog_color[3];
    uint32_t seed_lo, seed_hi;
    uint32_t n_region_pixels;
    uint32_t _pad;
};

// ---- ray queues (SoA of 16-byte groups, 56 B per ray) -------------------------------------
//   r0 = (origin.xyz, throughput)
//   r1 = (dir.xyz (normalised), bits(accumulator slot of the pixel))
//   r2 = (sample | depth << 16 | idcarrier << 24, path node index)
//   hit = (bits(toi), item index or -1, face id (+ n_tris for back faces), 0)
struct DRayQueue {
    float4* r0;
    float4* r1;
    uint2* r2;
    uint4* hit;
};

// shadow-ray queue, 48 B per ray
//   s0 = (origin.xyz, limit)      limit = distance to the light, or +FLT_MAX for directional lights
//   s1 = (dir.xyz, bits(receiver item | depth << 27))   the receiver's material alpha is looked up for occluded rays only
//   s2 = (contribution rgb, bits(accumulator slot))
// scenes whose top level has a packet form (rr_kernels.hip: beam_candidates)
#define RR_VIEW_NAN_BALLS 0x80000000u // DSceneView::compat
#define RR_BEAM_MAX_ITEMS 512u // 8 passes of 64 boxes: about what three steps of the per-ray walk cost
#define RR_BEAM_MIN_ITEMS 17u  // up to 16 items the tree is two levels: the per-ray walk is cheaper than the packet's set-up (shadow rays)
#ifndef RR_BEAM_MIN_ITEMS_CLOSEST
#define RR_BEAM_MIN_ITEMS_CLOSEST 17u // the same bound for closest-hit packets (their packet form also walks a mesh once per wave)
#endif

struct DShadowQueue {
    float4* s0;
    float4* s1;
    float4* s2;
};

// fixed-point accumulators: colour and normal scaled by 2^24, depth by 2^16
#define RR_FIX_SCALE 16777216.0f
#define RR_FIX_CLAMP 32768.0f
#define RR_DEPTH_SCALE 65536.0f
// Indexed by ACCUMULATOR SLOT: slots enumerate the region's pixels in trace order (8x8 blocks inside the
// tiles), so a wave of primary rays adds to 64 consecutive words of a plane.
struct DAccum {
    long long* rgb;    // 3 planes of n
    long long* normal; // 3 planes of n
    long long* depth;  // n
    uint32_t* object_id; // n
    unsigned long long n; // slots = pixels of the region
    uint32_t* flags;     // n: non-finite contributions seen by the pixel (RR_NF_*), so that k_resolve can give the sums the value
                         // the reference's f32 sums would have had (a NaN or +inf sample makes the channel 255, src/raytracing.rs:406-417)
};
enum : uint32_t { RR_NF_NAN = 1u, RR_NF_PINF = 1u << 3, RR_NF_NINF = 1u << 6,   // << channel (0 r, 1 g, 2 b)
                  RR_NF_DEPTH_NAN = 1u << 9, RR_NF_NORMAL_NAN = 1u << 10 };      // << component for the normal

// device-side counters (one block of 64-bit words)
enum {
    RR_CNT_PRIMARY = 0, RR_CNT_SECONDARY, RR_CNT_SHADOW, RR_CNT_SHADED,
    RR_CNT_WORDS = 8
};
