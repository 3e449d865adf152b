// rr_kernels.hip — the trace loop of rustray as a wavefront path tracer for gfx950.
//
// Replaces, for a whole frame, Raytracing::render / trace /
// get_color_depth_normal_id (reference src/raytracing.rs:275-998) and the
// Shape implementations under it (src/shape/mesh.rs, src/shape/sphere.rs,
// src/shape/mod.rs:510-629, :755-761).
//
// The reference recursion (every hit may spawn a reflection AND a refraction
// child, src/raytracing.rs:938-971) is flattened breadth-first: all path nodes
// of one recursion depth form a ray queue in HBM, and one depth level is
//     trace_closest  ->  shade  ->  trace_shadow
// `shade` appends the next level's rays (wave ballot + prefix compaction, one
// atomic per wave) and the shadow rays of its lights.  A node's colour is an
// affine function of its children (c = A + wR*R + wT*T with scalar weights), so
// each ray carries one scalar throughput and every node adds throughput * A into
// per-pixel fixed-point accumulators: integer adds commute, hence the frame is
// bit-identical for any scheduling, batching, tiling or GPU count.
//
// The trace kernels are persistent: a fixed grid of 256-thread workgroups walks
// the queue in 64-ray packets, half of them dealt round-robin, half pulled from a
// shared head; the queue length is read on the device.
#include "rr_device.h"
#include "rr_math.h"

#define RR_BLOCK 256
#define RR_SQ_SHARDS 32 // sub-queues of the shadow queue, one append counter each
#define RR_DEPTH_WIDE ((int)0x80000000) // k_shade: this lane's depth term does not fit its 32-bit sum (accum_depth_wide_merged)
#define RR_FIXED_SLOT_LIGHTS 32u // level 1 keeps fixed shadow slots for up to this many enabled lights (one bit per light in k_shade's sq_wrote)
#ifndef RR_SQ_STRIDE
#define RR_SQ_STRIDE 16 // words between two append counters: 64 B apart (packed into one line they cost k_shade 13-20 %)
#endif
#ifndef RR_TRACE_WAVES
#define RR_TRACE_WAVES 4 // waves per SIMD the trace kernels are built for (bounds VGPRs; LDS stack: RR_STACK_DEPTH KB per workgroup)
#endif
#ifndef RR_CLOSEST_WAVES
#define RR_CLOSEST_WAVES RR_TRACE_WAVES
#endif
#ifndef RR_SHADOW_WAVES
#define RR_SHADOW_WAVES RR_TRACE_WAVES
#endif
#define RR_WAVE 64
#ifndef RR_DYN_FETCH
#define RR_DYN_FETCH 4
#endif
// An item's REPORTED toi can lie in front of its box.  ray_toi_with_ball takes the root of b^2 - a c, which cancels
// catastrophically when the origin is far from the sphere: the discriminant of a grazing ray is rounding noise of the order
// u b^2, and the reported toi is off by up to sqrt(u) ~ 2.4e-4 of the distance (a sphere 2e4 units away "hit" 7 units in
// front of its box, by a ray that misses it: tools/fuzz_rays.py far, seed 419).  Wherever the top level prunes by distance
// -- against the best hit, or against the light -- the bound is therefore taken 1e-3 wider than the box distance says
// (and kept finite: the unused child slots of a node are boxes at infinity, which only a finite bound rejects).
#define RR_TOI_SLACK 1.001f
#ifndef RR_SHADOW_FIXED_STATIC_NUM
#define RR_SHADOW_FIXED_STATIC_NUM 7 // level 1 (fixed slots): sponza_syn shadow 6.3 -> 6.0 ms against one half
#define RR_SHADOW_FIXED_STATIC_DEN 8
#endif
#ifndef RR_SHADOW_STATIC_NUM
#define RR_SHADOW_STATIC_NUM 1
#define RR_SHADOW_STATIC_DEN 2
#endif

__constant__ float c_u8_to_f32[256]; // i / 255.0f, exactly as `(p[0] as f32) / 255.0`

// ---------------------------------------------------------------------------
// geometry primitives: parry3d 0.13 restated (ray_aabb.rs, ray_triangle.rs, ray_ball.rs)
// ---------------------------------------------------------------------------
struct LRay { f3 o, d; };

// ShapeBasics::get_inverse_ray, reference src/shape/mod.rs:755-761
RR_DEV LRay inverse_ray(const DItem& it, f3 o, f3 d, bool general_w) {
    LRay r;
    float ox = row4(it.inv0, o.x, o.y, o.z, 1.0f);
    float oy = row4(it.inv1, o.x, o.y, o.z, 1.0f);
    float oz = row4(it.inv2, o.x, o.y, o.z, 1.0f);
    if (general_w) { // Point3::from_homogeneous divides by w; w == 1 exactly for affine inverses
        float w = row4(it.inv3, o.x, o.y, o.z, 1.0f);
        ox = ox / w; oy = oy / w; oz = oz / w;
    }
    r.o = mk3(ox, oy, oz);
    r.d = mk3(row4(it.inv0, d.x, d.y, d.z, 0.0f), row4(it.inv1, d.x, d.y, d.z, 0.0f), row4(it.inv2, d.x, d.y, d.z, 0.0f));
    return r;
}
// (the w of an affine inverse is ((0 x + 0 y) + 0 z) + 1: exactly 1 for a finite point, NaN for any other -- 0 times an
// infinity -- so a non-finite point always takes the dividing form, and comes out NaN in every component as in the reference)
RR_DEV f3 to_local_point(const DItem& it, f3 p, bool general_w) {
    float x = row4(it.inv0, p.x, p.y, p.z, 1.0f);
    float y = row4(it.inv1, p.x, p.y, p.z, 1.0f);
    float z = row4(it.inv2, p.x, p.y, p.z, 1.0f);
    if (general_w || ((p.x - p.x) + (p.y - p.y)) + (p.z - p.z) != 0.0f) { float w = row4(it.inv3, p.x, p.y, p.z, 1.0f); x = x / w; y = y / w; z = z / w; }
    return mk3(x, y, z);
}
RR_DEV f3 to_world_normal(const DItem& it, f3 n) {
    return normalize3(mk3(row4(it.tr0, n.x, n.y, n.z, 0.0f), row4(it.tr1, n.x, n.y, n.z, 0.0f), row4(it.tr2, n.x, n.y, n.z, 0.0f)));
}

// Aabb::cast_local_ray(ray, f32::MAX, solid)
RR_DEV bool aabb_cast(const float* mins, const float* maxs, const LRay& ray, bool solid, float* toi) {
    float tmin = 0.0f, tmax = RR_FLT_MAX;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z};
    const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (d[i] == 0.0f) {
            if (o[i] < mins[i] || o[i] > maxs[i]) return false;
        } else {
            float denom = 1.0f / d[i];
            float a = (mins[i] - o[i]) * denom;
            float b = (maxs[i] - o[i]) * denom;
            float inear = (a > b) ? b : a;
            float ifar = (a > b) ? a : b;
            tmin = rs_max(tmin, inear);
            tmax = rs_min(tmax, ifar);
            if (tmin > tmax) return false;
        }
    }
    *toi = (tmin == 0.0f && !solid) ? tmax : tmin;
    return true;
}

// local_ray_intersection_with_triangle: toi and side only (the normal is rebuilt when shading).
// `back` is parry's FeatureId side (d >= 0); `neg` says the returned normal is -normalize(n) (t < 0).
// They differ only when the origin lies exactly in the triangle's plane.
// Written with a single exit: every arithmetic result is the same IEEE value as in parry's two branches
// (v = -ac.e | ac.e, w = ab.e | -ab.e, toi = -t/d | t/d; negation is exact), rejections keep parry's
// comparison forms so NaNs fall through exactly as they do there; the division runs for accepted hits only.
RR_DEV bool ray_triangle(f3 a, f3 ab, f3 ac, const LRay& ray, float* toi_out, uint32_t* side_out) {
    // ab = b - a, ac = c - a: computed once per triangle on the host (DTriX), with the IEEE sequence parry uses per ray
    const f3 n = cross3(ab, ac);
    const float d = dot3(n, ray.d);
    const f3 ap = ray.o - a;
    const float t = dot3(ap, n);
    const bool rej0 = (d == 0.0f) || (t < 0.0f && d < 0.0f) || (t > 0.0f && d > 0.0f);
    const bool back = !(d < 0.0f);
    const float dabs = rr_abs(d);
    const f3 e = -cross3(ray.d, ap);
    const float x = dot3(ac, e), y = dot3(ab, e);
    const bool neg = t < 0.0f;
    const float v = neg ? -x : x;
    const float w = neg ? y : -y;
    const bool rej1 = (v < 0.0f) || (v > dabs) || (w < 0.0f) || (v + w > dabs);
    if (rej0 || rej1) return false;
    const float invd = 1.0f / dabs;
    const float toi = (neg ? -t : t) * invd;
    if (!(toi <= RR_FLT_MAX)) return false;
    *toi_out = toi;
    *side_out = (back ? 2u : 0u) | (neg ? 1u : 0u);
    return true;
}

// ray_toi_with_ball + Ball::cast_local_ray_and_get_normal (centre = local origin)
RR_DEV bool ray_ball(float radius, const LRay& ray, bool solid, float* toi_out, bool* inside_out) {
    float a = dot3(ray.d, ray.d);
    float b = dot3(ray.o, ray.d);
    float c = dot3(ray.o, ray.o) - radius * radius;
    bool inside; float toi;
    if (a == 0.0f) {
        if (c > 0.0f) return false;
        inside = true; toi = 0.0f;
    } else if (c > 0.0f && b > 0.0f) {
        return false;
    } else {
        float delta = b * b - a * c;
        if (delta < 0.0f) return false;
        float sq = sqrtf(delta);
        float t = (-b - sq) / a;
        if (t <= 0.0f) { inside = true; toi = solid ? 0.0f : (-b + sq) / a; }
        else { inside = false; toi = t; }
    }
    if (toi > RR_FLT_MAX) return false;
    *toi_out = toi; *inside_out = inside;
    return true;
}

// ---------------------------------------------------------------------------
// BVH4 traversal (DNode4, rr_device.h).  Per-lane stack in LDS, lane-interleaved (conflict free), terminated by a
// sentinel entry instead of a depth test.
// ---------------------------------------------------------------------------
// A scene pointer is a GLOBAL pointer.  The trace kernels get the scene view as kernel arguments and the compiler knows;
// k_shade reads it from a device record (DShadeConst), where a pointer loaded from memory is generic and every access
// through it becomes a flat_load (aperture check, counted against both vmcnt and lgkmcnt).  The integer round trip gives
// the optimiser the address space back.
template <class T> RR_DEV const T* rr_global(const T* p) { return (const T*)(const __attribute__((address_space(1))) T*)(uintptr_t)p; }
#define STK(sp) s_stack[(sp) * RR_BLOCK + threadIdx.x]
#define RR_SENTINEL ((int)0x80000000) // bottom of every stack; root of an empty tree

// Developer instrumentation (-DRR_EXP_UTIL): active lanes per executed step, by kind.  Never in the shipped build.
#ifdef RR_EXP_UTIL
__device__ unsigned long long g_util[64];
__shared__ uint32_t s_util_kind; // 0: closest-hit level 1, 1: closest-hit deeper levels, 2: shadow rays (set by the kernels)
#define RR_UTIL(slot) { const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1) { \
        atomicAdd(&g_util[10 * s_util_kind + 2 * (slot)], (unsigned long long)__popcll(m_)); atomicAdd(&g_util[10 * s_util_kind + 2 * (slot) + 1], 1ull); } }
#define RR_UTIL_KIND(k) { s_util_kind = (k); __syncthreads(); }
#define RR_UTIL_NODE_SLOT (((const void*)nodes4_ptr_ == (const void*)sc.tnodes4 || (const void*)nodes4_ptr_ == (const void*)sc.tnodes4c) ? 0 : 2)
// steps whose address is the same in every active lane (g_util[30 + ...]: [0] same address, [1] same address and same key2)
#define RR_UTIL_UNI(slot, addr, key2) { const unsigned long long m_ = __ballot(1); const int l_ = __ffsll((long long)m_) - 1;                \
        const uint32_t a_ = (uint32_t)(addr), k_ = (uint32_t)(key2); const uint32_t ua_ = __shfl(a_, l_), uk_ = __shfl(k_, l_);               \
        const bool u1_ = __ballot(a_ == ua_) == m_; const bool u2_ = u1_ && __ballot(k_ == uk_) == m_;                                       \
        if ((int)(threadIdx.x & 63u) == l_) { if (u1_) atomicAdd(&g_util[30 + 10 * s_util_kind + 2 * (slot)], 1ull);                         \
                                              if (u2_) atomicAdd(&g_util[30 + 10 * s_util_kind + 2 * (slot) + 1], 1ull); } }
// node steps in which no lane has more than one (g_util[60]) / two (61) children hit, of all node steps (62)
#define RR_UTIL_ONE { const int nh_ = (int)h0 + (int)h1 + (int)h2 + (int)h3; const unsigned long long m_ = __ballot(1); const bool one_ = __ballot(nh_ > 1) == 0ull; const bool two_ = __ballot(nh_ > 2) == 0ull; \
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)m_) - 1) { if (one_) atomicAdd(&g_util[60], 1ull); if (two_) atomicAdd(&g_util[61], 1ull); atomicAdd(&g_util[62], 1ull); } }
#else
#define RR_UTIL(slot)
#define RR_UTIL_KIND(k)
#define RR_UTIL_UNI(slot, addr, key2)
#define RR_UTIL_ONE
#endif

// The traversal's own box test is NOT part of the parity contract (only the exact primitive tests decide
// hits), so its reciprocal is the hardware approximation.  The subtraction stays in front of the multiply:
// the fused form plane * inv - o * inv cancels catastrophically when the origin sits within the shadow bias
// of a box plane (measured as missed hits on scenes/spheres_room).
struct SlabRay { f3 o, inv; };
RR_DEV SlabRay make_slab(f3 o, f3 d) {
    SlabRay r; r.o = o;
    r.inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    return r;
}

// One BVH4 inner-node step: four slab tests, a five-exchange sorting network on (entry, child), the three
// farther children written far-to-near with the stack pointer advanced past the ones that were hit (a missed
// child sorts last and its slot is overwritten), and the nearest taken directly.  Single branch, like the BVH2 step.
#define RR_CSWAP(ka, ca, kb, cb) { const bool s_ = kb < ka; const float tk_ = s_ ? ka : kb; const int tc_ = s_ ? ca : cb; \
                                   ka = s_ ? kb : ka; ca = s_ ? cb : ca; kb = tk_; cb = tc_; }
// Per-walk constants of the 4-wide step: the ray in slab form, and for every axis which of the node's two plane
// rows is the near one for this ray's direction sign (row index 0/1), so that the step loads "near" and "far" rows
// directly instead of ordering the two plane distances of every child with a min and a max.
typedef float v2f __attribute__((ext_vector_type(2)));
// Rows are addressed as (uniform node array) + 32-bit byte offset, so the loads take the scalar-base form and the
// step needs one 32-bit add per row instead of 64-bit address arithmetic: off = (tree base + node) * 128 + row * 16.
struct Slab4 {
    f3 o, inv; uint32_t nx, fx, ny, fy, nz, fz, cc;
    // wave-uniform copies: `uni` when every lane that starts this walk has the same tree and the same direction signs, so
    // that a step whose node is the same in all of its lanes can fetch the rows ONCE through the scalar cache (u*: the same
    // row offsets in scalar registers)
    bool uni; uint32_t unx, ufx, uny, ufy, unz, ufz, ucc;
};
RR_DEV Slab4 make_slab4(const SlabRay& r, uint32_t node_base) {
    Slab4 s; s.o = r.o; s.inv = r.inv;
    const uint32_t sx = __float_as_uint(r.inv.x) >> 31, sy = __float_as_uint(r.inv.y) >> 31, sz = __float_as_uint(r.inv.z) >> 31;
    const uint32_t b = node_base << 7;
    s.nx = b + (sx << 4); s.fx = b + ((1u - sx) << 4);
    s.ny = b + ((2u + sy) << 4); s.fy = b + ((3u - sy) << 4);
    s.nz = b + ((4u + sz) << 4); s.fz = b + ((5u - sz) << 4);
    s.cc = b + (6u << 4);
#ifndef RR_NO_SCALAR_NODES
    const uint32_t key = b | (sx << 4) | (sy << 5) | (sz << 6); // b is a multiple of 128
    const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    s.uni = __ballot(key != ukey) == 0ull;
    const uint32_t ub = ukey & ~127u, ux = (ukey >> 4) & 1u, uy = (ukey >> 5) & 1u, uz = (ukey >> 6) & 1u;
    s.unx = ub + (ux << 4); s.ufx = ub + ((1u - ux) << 4);
    s.uny = ub + ((2u + uy) << 4); s.ufy = ub + ((3u - uy) << 4);
    s.unz = ub + ((4u + uz) << 4); s.ufz = ub + ((5u - uz) << 4);
    s.ucc = ub + (6u << 4);
#else
    s.uni = false; s.unx = s.ufx = s.uny = s.ufy = s.unz = s.ufz = s.ucc = 0u;
#endif
    return s;
}
RR_DEV DTriX tri_at(const DTriX* tris, uint32_t byte_off) { return *(const DTriX*)((const char*)tris + byte_off); }
RR_DEV float4 node_row(const DNode4* nodes, uint32_t byte_off) { return *(const float4*)((const char*)nodes + byte_off); }
// the same row at a wave-uniform offset, through the constant address space: one s_load_dwordx4 for the wave, the row
// arrives in scalar registers and feeds the packed subtracts directly.  (A vector load costs the L1 pipeline a quad of
// lanes per cycle whether or not the 64 addresses are equal: 16 cycles per row, and the walks are bound by exactly that.)
typedef float rr_f4v __attribute__((ext_vector_type(4)));
RR_DEV float4 node_row_uniform(const DNode4* nodes, uint32_t byte_off) {
    const rr_f4v v = *(const __attribute__((address_space(4))) rr_f4v*)((const __attribute__((address_space(4))) char*)(uintptr_t)nodes + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}
// (row - o) * inv for the four children of one plane row, as two packed pairs
#define RR_ROW(row, oc, ic, lo_, hi_) const v2f lo_ = (v2f{row.x, row.y} - v2f{oc, oc}) * v2f{ic, ic}; \
                                      const v2f hi_ = (v2f{row.z, row.w} - v2f{oc, oc}) * v2f{ic, ic};
// conservative hit test of one child from its three near and three far plane distances (same slack as slab2)
#define RR_CHILD(k_, h_, nx, ny, nz, fx, fy, fz)                                                               \
    float k_; bool h_;                                                                                         \
    {                                                                                                          \
        const float tn_ = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));                                               \
        const float tf_ = fminf(fminf(fx, fy), fminf(fz, RR_FLT_MAX));                                         \
        const float tc_ = tn_ * 0.999996f;                                                                     \
        h_ = tc_ <= tf_ * 1.000004f && tc_ <= bound_;                                                          \
        k_ = h_ ? tn_ : inf_;                                                                                  \
    }
#define RR_NODE4_ROWS_VECTOR(nodes4, s4)                                                                        \
        const uint32_t no_ = (uint32_t)cur << 7;                                                               \
        const float4 rnx = node_row(nodes4, no_ + (s4).nx), rfx = node_row(nodes4, no_ + (s4).fx);             \
        const float4 rny = node_row(nodes4, no_ + (s4).ny), rfy = node_row(nodes4, no_ + (s4).fy);             \
        const float4 rnz = node_row(nodes4, no_ + (s4).nz), rfz = node_row(nodes4, no_ + (s4).fz);             \
        const float4 cc = node_row(nodes4, no_ + (s4).cc);
#define RR_NODE4_ROWS_UNIFORM(nodes4, s4)                                                                       \
        const uint32_t no_ = (uint32_t)ucur_ << 7;                                                             \
        const float4 rnx = node_row_uniform(nodes4, no_ + (s4).unx), rfx = node_row_uniform(nodes4, no_ + (s4).ufx); \
        const float4 rny = node_row_uniform(nodes4, no_ + (s4).uny), rfy = node_row_uniform(nodes4, no_ + (s4).ufy); \
        const float4 rnz = node_row_uniform(nodes4, no_ + (s4).unz), rfz = node_row_uniform(nodes4, no_ + (s4).ufz); \
        const float4 cc = node_row_uniform(nodes4, no_ + (s4).ucc);
#define RR_NODE4_TESTS(s4, bound)                                                                              \
        const float bound_ = (bound);                                                                          \
        const float inf_ = __builtin_inff();                                                                   \
        RR_ROW(rnx, (s4).o.x, (s4).inv.x, nx01, nx23) RR_ROW(rfx, (s4).o.x, (s4).inv.x, fx01, fx23)            \
        RR_ROW(rny, (s4).o.y, (s4).inv.y, ny01, ny23) RR_ROW(rfy, (s4).o.y, (s4).inv.y, fy01, fy23)            \
        RR_ROW(rnz, (s4).o.z, (s4).inv.z, nz01, nz23) RR_ROW(rfz, (s4).o.z, (s4).inv.z, fz01, fz23)            \
        RR_CHILD(k0, h0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)                                       \
        RR_CHILD(k1, h1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)                                       \
        RR_CHILD(k2, h2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)                                       \
        RR_CHILD(k3, h3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y) RR_UTIL_ONE
// Two thirds of the node steps of the contract frame have at most ONE child hit in every lane (9.6 % have more than two):
// then nothing is ordered and nothing is pushed.  The test is scalar (the hit flags are lane masks).  Closest-hit walks
// only (-1 % sponza_syn, -3 % lotus_syn): the shadow kernel, at its register limit, loses 3 % to it.
#define RR_NODE4_SINGLE_HIT                                                                                    \
        const bool multi_ = (h0 && (h1 || h2 || h3)) || (h1 && (h2 || h3)) || (h2 && h3);                      \
        if (__ballot(multi_) == 0ull) {                                                                        \
            const int c_ = __float_as_int(h0 ? cc.x : (h1 ? cc.y : (h2 ? cc.z : cc.w)));                       \
            if (h0 || h1 || h2 || h3) cur = c_;                                                                \
            else { sp--; cur = STK(sp); }                                                                      \
        } else
#define RR_NODE4_DESCEND_SORTED                                                                                \
        RR_NODE4_SINGLE_HIT {                                                                                  \
        int c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w); \
        RR_CSWAP(k0, c0, k1, c1) RR_CSWAP(k2, c2, k3, c3) RR_CSWAP(k0, c0, k2, c2) RR_CSWAP(k1, c1, k3, c3) RR_CSWAP(k1, c1, k2, c2) \
        STK(sp) = c3; sp += (k3 < inf_) ? 1 : 0;                                                               \
        STK(sp) = c2; sp += (k2 < inf_) ? 1 : 0;                                                               \
        STK(sp) = c1; sp += (k1 < inf_) ? 1 : 0;                                                               \
        if (k0 < inf_) cur = c0;                                                                               \
        else { sp--; cur = STK(sp); }                                                                          \
        }
#define RR_NODE4_DESCEND_SORTED_PLAIN                                                                          \
        {                                                                                                      \
        int c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w); \
        RR_CSWAP(k0, c0, k1, c1) RR_CSWAP(k2, c2, k3, c3) RR_CSWAP(k0, c0, k2, c2) RR_CSWAP(k1, c1, k3, c3) RR_CSWAP(k1, c1, k2, c2) \
        STK(sp) = c3; sp += (k3 < inf_) ? 1 : 0;                                                               \
        STK(sp) = c2; sp += (k2 < inf_) ? 1 : 0;                                                               \
        STK(sp) = c1; sp += (k1 < inf_) ? 1 : 0;                                                               \
        if (k0 < inf_) cur = c0;                                                                               \
        else { sp--; cur = STK(sp); }                                                                          \
        }
#define RR_NODE4_DESCEND_ANY                                                                                   \
        (void)k0; (void)k1; (void)k2; (void)k3;                                                                \
        STK(sp) = __float_as_int(cc.w); sp += h3 ? 1 : 0;                                                      \
        STK(sp) = __float_as_int(cc.z); sp += h2 ? 1 : 0;                                                      \
        STK(sp) = __float_as_int(cc.y); sp += h1 ? 1 : 0;                                                      \
        if (h0) cur = __float_as_int(cc.x);                                                                    \
        else { sp--; cur = STK(sp); }
// A step whose node is the same in all of its lanes (on a walk that is `uni`) takes the scalar form of the loads.
#ifndef RR_NO_SCALAR_NODES
#define RR_NODE4_FORM(nodes4, s4, bound, DESCEND)                                                              \
    {                                                                                                          \
        const void* nodes4_ptr_ = (nodes4); (void)nodes4_ptr_;                                                 \
        RR_UTIL(RR_UTIL_NODE_SLOT) RR_UTIL_UNI(RR_UTIL_NODE_SLOT, cur, ((s4).nx & 16u) | ((s4).ny & 16u) << 1 | ((s4).nz & 16u) << 2 | ((s4).cc << 3)) \
        const int ucur_ = __builtin_amdgcn_readfirstlane(cur);                                                 \
        if ((s4).uni && __ballot(cur != ucur_) == 0ull) { RR_NODE4_ROWS_UNIFORM(nodes4, s4) RR_NODE4_TESTS(s4, bound) DESCEND } \
        else { RR_NODE4_ROWS_VECTOR(nodes4, s4) RR_NODE4_TESTS(s4, bound) DESCEND }                            \
    }
#else
#define RR_NODE4_FORM(nodes4, s4, bound, DESCEND)                                                              \
    {                                                                                                          \
        const void* nodes4_ptr_ = (nodes4); (void)nodes4_ptr_;                                                 \
        RR_UTIL(RR_UTIL_NODE_SLOT) RR_UTIL_UNI(RR_UTIL_NODE_SLOT, cur, ((s4).nx & 16u) | ((s4).ny & 16u) << 1 | ((s4).nz & 16u) << 2 | ((s4).cc << 3)) \
        RR_NODE4_ROWS_VECTOR(nodes4, s4) RR_NODE4_TESTS(s4, bound) DESCEND                                     \
    }
#endif
#define RR_NODE4_STEP(nodes4, s4, bound) RR_NODE4_FORM(nodes4, s4, bound, RR_NODE4_DESCEND_SORTED)
#define RR_NODE4_STEP_PLAIN(nodes4, s4, bound) RR_NODE4_FORM(nodes4, s4, bound, RR_NODE4_DESCEND_SORTED_PLAIN)
// The same step for walks that only ask whether anything is hit (shadow queries inside one mesh): the order in which
// the children are visited does not matter, so the hit children are pushed in slot order and the sort is skipped.
#define RR_NODE4_STEP_ANY(nodes4, s4, bound) RR_NODE4_FORM(nodes4, s4, bound, RR_NODE4_DESCEND_ANY)
#define RR_BLAS_NODES(sc, it) ((sc).nodes4) // uniform; the tree's base is folded into the node offsets of the Slab4
#define RR_BLAS_ROOT(it) ((it).root4)
#define RR_BLAS_STEP(nodes, sr, bound) RR_NODE4_STEP(nodes, sr, bound)
#define RR_BLAS_STEP_ANY(nodes, sr, bound) RR_NODE4_STEP_ANY(nodes, sr, bound)
#define RR_BLAS_SLAB(r) make_slab4(make_slab((r).o, (r).d), it.node_base4)
typedef DNode4 BlasNode;
typedef Slab4 BlasSlab;

// Nearest triangle of one mesh (TriMesh::cast_local_ray_and_get_normal,
// reference src/shape/mesh.rs:67).  Ties at bit-equal toi go to the lowest
// ORIGINAL face index.  `gbound`: hits beyond it cannot win upstream.
// Returns slot (leaf-order triangle index) and side.
struct TriBest { float t; uint32_t slot; uint32_t face; uint32_t side; bool found; };

// Postponed leaves (RR_POSTPONE): a lane that reaches a leaf parks it and keeps walking; the wave tests parked
// leaves together once RR_PEND_NUM/RR_PEND_DEN of its unfinished lanes hold one, or nobody can walk on.  The
// order in which triangles are tested is free: the winner is the minimum over (toi, face) and the walk only ever
// prunes with a bound no smaller than the current best.  (Measured before: node steps ran with ~27 of 64 lanes,
// triangle tests with 7-15.)
#ifndef RR_PEND_NUM
#define RR_PEND_NUM 2
#define RR_PEND_DEN 3
#endif

#define RR_TRI_CLOSEST(tr, slot_)                                                                               \
            {                                                                                                  \
                float t; uint32_t side;                                                                        \
                if (ray_triangle(mk3(tr.t0.x, tr.t0.y, tr.t0.z), mk3(tr.t1.x, tr.t1.y, tr.t1.z),               \
                                 mk3(tr.t1.w, tr.t2.x, tr.t2.y), ray, &t, &side)) {                            \
                    const uint32_t face = __float_as_uint(tr.t0.w);                                            \
                    /* (best starts at (FLT_MAX, face 0xffffffff): the first hit always wins without asking best.found) */ \
                    if (t < best.t || (t == best.t && face < best.face)) {                                     \
                        best.found = true; best.t = t; best.slot = (slot_); best.face = face; best.side = side; \
                    }                                                                                          \
                }                                                                                              \
            }
// A leaf that is the same in every lane of a walk that shares its tree (three quarters of the triangle tests of level 1)
// is fetched through the scalar cache, TWO triangles per wait: the tests of a leaf are a chain of load -> test -> load.
#define RR_TRI_FETCH(t_, o_) t_.t0 = node_row_uniform((const DNode4*)sc.trix, o_); t_.t1 = node_row_uniform((const DNode4*)sc.trix, (o_) + 16u); t_.t2 = node_row_uniform((const DNode4*)sc.trix, (o_) + 32u);
#ifndef RR_NO_SCALAR_LEAVES
#define RR_LEAF_CLOSEST(leaf)                                                                                  \
    {                                                                                                          \
        const int uleaf_ = __builtin_amdgcn_readfirstlane(leaf);                                               \
        const uint32_t utri_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)tri_base_);                       \
        /* (the triangle base is compared too: meshes small enough to be ONE leaf add no nodes and share a node base) */ \
        if (SCALAR_LEAVES && sr.uni && __ballot((leaf) != uleaf_ || tri_base_ != utri_) == 0ull) {                              \
            const uint32_t ucode = (uint32_t)~uleaf_;                                                          \
            const uint32_t ufirst = RR_LEAF_FIRST(ucode), ucount = RR_LEAF_COUNT(ucode);                       \
            const uint32_t ubase = utri_ + ufirst;                                                             \
            for (uint32_t i = 0; i < ucount; i += 2u) {                                                        \
                RR_UTIL(3)                                                                                     \
                const bool two_ = i + 1u < ucount;                                                             \
                const uint32_t o0 = (ubase + i) * 48u, o1 = (ubase + i + (two_ ? 1u : 0u)) * 48u;              \
                DTriX ta, tb;                                                                                  \
                RR_TRI_FETCH(ta, o0) RR_TRI_FETCH(tb, o1)                                                      \
                RR_TRI_CLOSEST(ta, ufirst + i)                                                                 \
                if (two_) RR_TRI_CLOSEST(tb, ufirst + i + 1u)                                                  \
            }                                                                                                  \
        } else {                                                                                               \
            const uint32_t code = (uint32_t)~(leaf);                                                           \
            const uint32_t first = RR_LEAF_FIRST(code), count = RR_LEAF_COUNT(code);                           \
            for (uint32_t i = 0; i < count; i++) {                                                             \
                RR_UTIL(3) RR_UTIL_UNI(3, tri_base_ + first + i, 0)                                            \
                const DTriX tr = tri_at(sc.trix, (tri_base_ + first + i) * 48u);                               \
                RR_TRI_CLOSEST(tr, first + i)                                                                  \
            }                                                                                                  \
        }                                                                                                      \
    }
#else
#define RR_LEAF_CLOSEST(leaf)                                                                                  \
    {                                                                                                          \
        const uint32_t code = (uint32_t)~(leaf);                                                               \
        const uint32_t first = RR_LEAF_FIRST(code), count = RR_LEAF_COUNT(code);                               \
        for (uint32_t i = 0; i < count; i++) {                                                                 \
            RR_UTIL(3) RR_UTIL_UNI(3, tri_base_ + first + i, 0)                                                \
            const DTriX tr = tri_at(sc.trix, (tri_base_ + first + i) * 48u);                                   \
            RR_TRI_CLOSEST(tr, first + i)                                                                      \
        }                                                                                                      \
    }
#endif
#define RR_LEAF_ANY(leaf)                                                                                      \
    {                                                                                                          \
        const uint32_t code = (uint32_t)~(leaf);                                                               \
        const uint32_t first = RR_LEAF_FIRST(code), count = RR_LEAF_COUNT(code);                               \
        for (uint32_t i = 0; i < count; i++) {                                                                 \
            RR_UTIL(3) RR_UTIL_UNI(3, tri_base_ + first + i, 0)                                                \
            const DTriX tr = tri_at(sc.trix, (tri_base_ + first + i) * 48u);                                   \
            float t; uint32_t side;                                                                            \
            if (ray_triangle(mk3(tr.t0.x, tr.t0.y, tr.t0.z), mk3(tr.t1.x, tr.t1.y, tr.t1.z),                   \
                             mk3(tr.t1.w, tr.t2.x, tr.t2.y), ray, &t, &side)) {                                \
                any = true;                                                                                    \
                if (t <= limit) within = true;                                                                 \
            }                                                                                                  \
        }                                                                                                      \
    }

// SCALAR_LEAVES: the closest-hit kernels' form (see RR_LEAF_CLOSEST); the shadow kernel, at its register limit, keeps the plain loop.
template <bool SCALAR_LEAVES>
RR_DEV void blas_closest(const DSceneView& sc, const DItem& it, const LRay& ray, float gbound,
                         int* s_stack, int sp_base, TriBest* out) {
    TriBest best; best.found = false; best.t = RR_FLT_MAX; best.slot = 0; best.face = 0xffffffffu; best.side = 0u;
    const BlasSlab sr = RR_BLAS_SLAB(ray);
    const BlasNode* nodes = RR_BLAS_NODES(sc, it);
    const uint32_t tri_base_ = it.tri_base; // triangles, like node rows, are addressed as uniform base + 32-bit offset
    int sp = sp_base;
    STK(sp) = RR_SENTINEL; sp++;
    int cur = RR_BLAS_ROOT(it);
    RR_UTIL(4)
    int pend = 0; // parked leaf (leaf codes are negative), 0 = none
    for (;;) {
        if (cur >= 0) {
            RR_BLAS_STEP(nodes, sr, fminf(gbound, best.t))
        } else if (pend == 0 && cur != RR_SENTINEL) {
            pend = cur; sp--; cur = STK(sp);
        }
        const unsigned long long can_walk = __ballot(cur >= 0 || (pend == 0 && cur != RR_SENTINEL));
        const unsigned long long parked = __ballot(pend != 0);
        if ((can_walk | parked) == 0ull) break; // every lane of this walk is done
        const unsigned long long alive = __ballot(cur != RR_SENTINEL || pend != 0);
        if (can_walk == 0ull || __popcll(parked) * RR_PEND_DEN >= __popcll(alive) * RR_PEND_NUM) {
            if (pend != 0) { RR_LEAF_CLOSEST(pend) pend = 0; }
        }
    }
    *out = best;
}

// Shadow query of one mesh: is there ANY hit, and is there one with toi <= limit?
// Stops at the first hit within the limit.
RR_DEV void blas_any(const DSceneView& sc, const DItem& it, const LRay& ray, float limit,
                     int* s_stack, int sp_base, bool* found_any, bool* found_within) {
    bool any = false, within = false;
    const BlasSlab sr = RR_BLAS_SLAB(ray);
    const BlasNode* nodes = RR_BLAS_NODES(sc, it);
    const uint32_t tri_base_ = it.tri_base; // triangles, like node rows, are addressed as uniform base + 32-bit offset
    int sp = sp_base;
    STK(sp) = RR_SENTINEL; sp++;
    int cur = RR_BLAS_ROOT(it);
    RR_UTIL(4)
    // until some hit is known every box matters; afterwards only boxes that can still hold a hit within the limit
    int pend = 0;
    for (;;) {
        if (cur >= 0) {
            RR_BLAS_STEP_ANY(nodes, sr, any ? limit : RR_FLT_MAX)
        } else if (pend == 0 && cur != RR_SENTINEL) {
            pend = cur; sp--; cur = STK(sp);
        }
        const unsigned long long can_walk = __ballot(cur >= 0 || (pend == 0 && cur != RR_SENTINEL));
        const unsigned long long parked = __ballot(pend != 0);
        if ((can_walk | parked) == 0ull) break;
        const unsigned long long alive = __ballot(cur != RR_SENTINEL || pend != 0);
        if (can_walk == 0ull || __popcll(parked) * RR_PEND_DEN >= __popcll(alive) * RR_PEND_NUM) {
            if (pend != 0) {
                RR_LEAF_ANY(pend)
                pend = 0;
                if (within) cur = RR_SENTINEL; // decided: this lane stops walking
            }
        }
    }
    *found_any = any; *found_within = within;
}

// ---------------------------------------------------------------------------
// The per-mesh walk of a PACKET: all 64 lanes walk ONE mesh with ONE wave-uniform control flow (trace_closest_packet /
// trace_shadow_packet visit a candidate item with every lane together).  A node step costs its instructions per WAVE, not per
// lane, and in a packet of 64 samples of one pixel three quarters of the per-lane steps had the same node in every lane anyway:
// here the node index, the stack (one LDS word per entry, the wave's own column) and the order in which children are tried are
// scalar; each lane still tests the four child boxes with ITS ray and its own bound, and tests a leaf's triangles only if ITS
// box test of that leaf passed -- so a lane's set of tested triangles is what its own walk would test, up to nodes that a bound
// (its best hit so far) prunes, which never changes a result: the frame is the same bit for bit.  What goes away per step is the
// per-lane bookkeeping: the five-exchange sorting network, three LDS pushes with their addresses, the parked-leaf ballots.
// Leaves are tested when their parent is visited (under the lanes' hit flags of that step); only inner nodes are stacked, in
// the order of the FIRST hitting lane's entry distances (any order is correct; near-first prunes best).
// (A bound prunes by the ORDER in which hits are found, and the order here follows the first hitting lane.  That never matters for
// a triangle hit that lies inside its leaf's box; the per-lane walks have the same dependence on their wave through the moment at
// which parked leaves are tested.  Where a reported toi lies in front of the leaf's box -- rounding noise for origins >~ 1e4 mesh
// sizes away, DESIGN.md D12 -- neither form promises the reference's pick.)
// `in`: this lane takes part (its exact test of the item's box passed).  Must be called by all 64 lanes.
// ---------------------------------------------------------------------------
#define RR_PK_STK(sp_) s_stack[(sp_) * RR_BLOCK + wave_col_]
// the triangles of a wave-uniform leaf for the lanes with `hit_`, two per wait through the scalar cache
#define RR_PK_LEAF(code_, hit_, TEST)                                                                          \
    {                                                                                                          \
        const uint32_t ucode = (uint32_t)~(code_);                                                             \
        const uint32_t ufirst = RR_LEAF_FIRST(ucode), ucount = RR_LEAF_COUNT(ucode);                           \
        const uint32_t ubase = utri_ + ufirst;                                                                 \
        for (uint32_t i = 0; i < ucount; i += 2u) {                                                            \
            const bool two_ = i + 1u < ucount;                                                                 \
            const uint32_t o0 = (ubase + i) * 48u, o1 = (ubase + i + (two_ ? 1u : 0u)) * 48u;                  \
            DTriX ta, tb;                                                                                      \
            RR_TRI_FETCH(ta, o0) RR_TRI_FETCH(tb, o1)                                                          \
            if (hit_) { TEST(ta, ufirst + i) if (two_) TEST(tb, ufirst + i + 1u) }                             \
        }                                                                                                      \
    }
#define RR_TRI_ANY(tr, slot_)                                                                                   \
            {                                                                                                  \
                float t; uint32_t side;                                                                        \
                if (ray_triangle(mk3(tr.t0.x, tr.t0.y, tr.t0.z), mk3(tr.t1.x, tr.t1.y, tr.t1.z),               \
                                 mk3(tr.t1.w, tr.t2.x, tr.t2.y), ray, &t, &side)) {                            \
                    any = true;                                                                                \
                    if (t <= limit) within = true;                                                             \
                }                                                                                              \
            }
// the child test of RR_CHILD with ONE compare (entry <= min(exit, bound): the same predicate; a lane mask less to combine on the
// scalar unit, which the packet walk leans on) and the raw entry distance as the key (only hit children's keys are read)
#define RR_PK_CHILD(k_, h_, nx, ny, nz, fx, fy, fz)                                                            \
        {                                                                                                      \
            const float tn_ = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));                                           \
            const float tf_ = fminf(fminf(fx, fy), fminf(fz, RR_FLT_MAX));                                     \
            h_ = tn_ * 0.999996f <= fminf(tf_ * 1.000004f, bound_);                                            \
            k_ = tn_;                                                                                          \
        }
// one uniform node: tests, leaves, and the choice of the next node.  BOUND: the lane's pruning bound; TEST: the triangle macro;
// LIVE: the lane still wants hits (any-hit walks drop a lane once it is decided)
#define RR_PK_NODE(BOUND, TEST, LIVE)                                                                          \
    {                                                                                                          \
        const int ucur_ = cur;                                                                                 \
        RR_NODE4_ROWS_UNIFORM(nodes, sr)                                                                       \
        bool h0 = false, h1 = false, h2 = false, h3 = false;                                                   \
        float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f, k3 = 0.0f;                                                      \
        if (LIVE) {                                                                                            \
            const float bound_ = (BOUND);                                                                      \
            RR_ROW(rnx, sr.o.x, sr.inv.x, nx01, nx23) RR_ROW(rfx, sr.o.x, sr.inv.x, fx01, fx23)                \
            RR_ROW(rny, sr.o.y, sr.inv.y, ny01, ny23) RR_ROW(rfy, sr.o.y, sr.inv.y, fy01, fy23)                \
            RR_ROW(rnz, sr.o.z, sr.inv.z, nz01, nz23) RR_ROW(rfz, sr.o.z, sr.inv.z, fz01, fz23)                \
            RR_PK_CHILD(k0, h0, nx01.x, ny01.x, nz01.x, fx01.x, fy01.x, fz01.x)                                \
            RR_PK_CHILD(k1, h1, nx01.y, ny01.y, nz01.y, fx01.y, fy01.y, fz01.y)                                \
            RR_PK_CHILD(k2, h2, nx23.x, ny23.x, nz23.x, fx23.x, fy23.x, fz23.x)                                \
            RR_PK_CHILD(k3, h3, nx23.y, ny23.y, nz23.y, fx23.y, fy23.y, fz23.y)                                \
        }                                                                                                      \
        const int c0 = __float_as_int(cc.x), c1 = __float_as_int(cc.y), c2 = __float_as_int(cc.z), c3 = __float_as_int(cc.w); \
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);   \
        /* leaves of this node: tested now, by the lanes that hit them */                                      \
        if (m0 != 0ull && c0 < 0) RR_PK_LEAF(c0, h0, TEST)                                                     \
        if (m1 != 0ull && c1 < 0) RR_PK_LEAF(c1, h1, TEST)                                                     \
        if (m2 != 0ull && c2 < 0) RR_PK_LEAF(c2, h2, TEST)                                                     \
        if (m3 != 0ull && c3 < 0) RR_PK_LEAF(c3, h3, TEST)                                                     \
        /* inner children that some lane hits: none or one (most steps) needs no order; otherwise they are keyed by the entry distance */ \
        /* of the first lane that hits them (non-negative floats order as integers) and sorted in scalar registers */ \
        const bool i0 = m0 != 0ull && c0 >= 0, i1 = m1 != 0ull && c1 >= 0, i2 = m2 != 0ull && c2 >= 0, i3 = m3 != 0ull && c3 >= 0; \
        const int n_in = (int)i0 + (int)i1 + (int)i2 + (int)i3;                                                \
        if (n_in == 1) cur = i0 ? c0 : (i1 ? c1 : (i2 ? c2 : c3));                                             \
        else if (n_in > 1) {                                                                                   \
            uint32_t q0 = 0xffffffffu, q1 = 0xffffffffu, q2 = 0xffffffffu, q3 = 0xffffffffu;                   \
            if (i0) q0 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(k0), __ffsll((long long)m0) - 1);  \
            if (i1) q1 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(k1), __ffsll((long long)m1) - 1);  \
            if (i2) q2 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(k2), __ffsll((long long)m2) - 1);  \
            if (i3) q3 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(k3), __ffsll((long long)m3) - 1);  \
            int e0 = c0, e1 = c1, e2 = c2, e3 = c3;                                                            \
            RR_SCSWAP(q0, e0, q1, e1) RR_SCSWAP(q2, e2, q3, e3) RR_SCSWAP(q0, e0, q2, e2) RR_SCSWAP(q1, e1, q3, e3) RR_SCSWAP(q1, e1, q2, e2) \
            if (q3 != 0xffffffffu) { if (lane_ == 0u) RR_PK_STK(sp) = e3; sp++; }                               \
            if (q2 != 0xffffffffu) { if (lane_ == 0u) RR_PK_STK(sp) = e2; sp++; }                               \
            if (lane_ == 0u) RR_PK_STK(sp) = e1;                                                               \
            sp++;                                                                                              \
            cur = e0;                                                                                          \
        }                                                                                                      \
        else if (sp > sp_base) { sp--; __builtin_amdgcn_wave_barrier(); cur = __builtin_amdgcn_readfirstlane(RR_PK_STK(sp)); } \
        else cur = RR_SENTINEL;                                                                                \
    }
#define RR_SCSWAP(ka, ca, kb, cb) { const bool s_ = kb < ka; const uint32_t tk_ = s_ ? ka : kb; const int tc_ = s_ ? ca : cb; \
                                    ka = s_ ? kb : ka; ca = s_ ? cb : ca; kb = tk_; cb = tc_; }

// returns false (nothing done) when the lanes do not share their direction signs in the mesh's space: the caller walks per lane
RR_DEV bool blas_closest_packet(const DSceneView& sc, const DItem& it, const LRay& ray, bool in, float gbound,
                                int* s_stack, int sp_base, TriBest* out) {
    const BlasSlab sr = RR_BLAS_SLAB(ray);
    if (!sr.uni) return false;
    TriBest best; best.found = false; best.t = RR_FLT_MAX; best.slot = 0; best.face = 0xffffffffu; best.side = 0u;
    const BlasNode* nodes = RR_BLAS_NODES(sc, it);
    const uint32_t utri_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tri_base);
    const uint32_t wave_col_ = threadIdx.x & ~(RR_WAVE - 1u), lane_ = threadIdx.x & (RR_WAVE - 1u);
    int sp = sp_base;
    int cur = __builtin_amdgcn_readfirstlane(RR_BLAS_ROOT(it));
    if (cur < 0 && cur != RR_SENTINEL) { RR_PK_LEAF(cur, in, RR_TRI_CLOSEST) cur = RR_SENTINEL; } // a mesh of one leaf
    while (cur >= 0) RR_PK_NODE(fminf(gbound, best.t), RR_TRI_CLOSEST, in)
    *out = best;
    return true;
}
RR_DEV bool blas_any_packet(const DSceneView& sc, const DItem& it, const LRay& ray, bool in, float limit,
                            int* s_stack, int sp_base, bool* found_any, bool* found_within) {
    const BlasSlab sr = RR_BLAS_SLAB(ray);
    if (!sr.uni) return false;
    bool any = false, within = false;
    const BlasNode* nodes = RR_BLAS_NODES(sc, it);
    const uint32_t utri_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tri_base);
    const uint32_t wave_col_ = threadIdx.x & ~(RR_WAVE - 1u), lane_ = threadIdx.x & (RR_WAVE - 1u);
    int sp = sp_base;
    int cur = __builtin_amdgcn_readfirstlane(RR_BLAS_ROOT(it));
    if (cur < 0 && cur != RR_SENTINEL) { RR_PK_LEAF(cur, in, RR_TRI_ANY) cur = RR_SENTINEL; }
    // a lane is decided once it has a hit within the limit; until some hit is known every box matters, afterwards only boxes
    // that can still hold a hit within the limit (as in blas_any); the walk ends when no lane is left
    while (cur >= 0) {
        RR_PK_NODE(any ? limit : RR_FLT_MAX, RR_TRI_ANY, in && !within)
        if (__ballot(in && !within) == 0ull) break;
    }
    *found_any = any; *found_within = within;
    return true;
}

// ---------------------------------------------------------------------------
// Raytracing::trace (reference src/raytracing.rs:429-490) per item
// ---------------------------------------------------------------------------
// Aabb::cast_local_ray with the entry distance kept beside the returned toi: origin inside a
// non-solid box returns the EXIT distance as toi (the sort key) although hits may be nearer.
RR_DEV bool aabb_cast2(const float* mins, const float* maxs, const LRay& ray, bool solid, float* toi, float* tmin_out) {
    float tmin = 0.0f, tmax = RR_FLT_MAX;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z};
    const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (d[i] == 0.0f) {
            if (o[i] < mins[i] || o[i] > maxs[i]) return false;
        } else {
            float denom = 1.0f / d[i];
            float a = (mins[i] - o[i]) * denom;
            float b = (maxs[i] - o[i]) * denom;
            float inear = (a > b) ? b : a;
            float ifar = (a > b) ? a : b;
            tmin = rs_max(tmin, inear);
            tmax = rs_min(tmax, ifar);
            if (tmin > tmax) return false;
        }
    }
    *toi = (tmin == 0.0f && !solid) ? tmax : tmin;
    *tmin_out = tmin;
    return true;
}

// candidate filter of :454 on the texture-less material cache
RR_DEV bool item_passes(uint32_t flags, bool for_shadow, uint32_t depth) {
    if (!(flags & RR_IF_VISIBLE)) return false;
    if (!(flags & RR_IF_CACHE_ALPHA_POS)) return false;
    if (for_shadow && !(flags & RR_IF_CACHE_CAST_SHADOW)) return false;
    if ((flags & RR_IF_CACHE_REFL_ONLY) && !(depth > 1u)) return false;
    return true;
}

struct Closest { float t; int item; uint32_t face; float key; bool found; bool nan_seen; }; // nan_seen: a ball answered Some(NaN) (trace_closest_ordered)

// The reference sorts candidates by bbox distance (stable) and keeps strictly
// smaller toi, so among equal toi the smaller (bbox distance, item index) wins.
RR_DEV void closest_item(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth,
                         int* s_stack, int sp_base, Closest* best) {
    RR_UTIL(1)
    const DItem& it = rr_global(sc.items)[idx];
    uint32_t flags = it.flags;
    if (!item_passes(flags, false, depth)) return;
    LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
    bool solid = (flags & RR_IF_SOLID_BASE) != 0u;
    float key;
    if (!aabb_cast(it.bmin, it.bmax, lr, solid, &key)) return;
    if (key != key) return; // NaN distance: treated as a miss (the reference panics)
    float t; uint32_t face;
    if (flags & RR_IF_SPHERE) {
        bool inside;
        if (!ray_ball(it.radius, lr, solid, &t, &inside)) return;
        if (t != t) { best->nan_seen = true; return; } // Some(NaN): what it does to the result depends on the candidate ORDER (trace_closest_ordered)
        face = 0u;
    } else {
        if (it.n_tris == 0u) return;
        TriBest tb;
        blas_closest<true>(sc, it, lr, best->found ? best->t : RR_FLT_MAX, s_stack, sp_base, &tb);
        if (!tb.found) return;
        t = tb.t;
        face = tb.slot | (tb.side << 30); // bit31 back face, bit30 negated normal
    }
    bool better = !best->found || t < best->t ||
                  (t == best->t && (key < best->key || (key == best->key && idx < best->item)));
    if (better) { best->found = true; best->t = t; best->item = idx; best->face = face; best->key = key; }
}

// closest_item for a candidate that ALL lanes of a packet visit together (trace_closest_packet): a mesh is walked once per wave
// (blas_closest_packet), with the lanes whose exact box test passed taking part.  Same result as closest_item, lane by lane.
#ifndef RR_NO_PACKET_WALK
RR_DEV void closest_item_packet(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, int* s_stack, Closest* best) {
    const DItem& it = rr_global(sc.items)[idx];
    const uint32_t flags = it.flags;
    if (flags & RR_IF_SPHERE) { closest_item(sc, idx, o, d, depth, s_stack, 0, best); return; } // (wave-uniform: one item)
    RR_UTIL(1)
    bool in = item_passes(flags, false, depth) && it.n_tris != 0u;
    const LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
    float key = 0.0f;
    in = in && aabb_cast(it.bmin, it.bmax, lr, (flags & RR_IF_SOLID_BASE) != 0u, &key);
    in = in && key == key; // NaN distance: treated as a miss (the reference panics)
    if (__ballot(in) == 0ull) return;
    const float gbound = best->found ? best->t : RR_FLT_MAX;
    TriBest tb; tb.found = false; tb.t = RR_FLT_MAX; tb.slot = 0u; tb.face = 0xffffffffu; tb.side = 0u;
    if (!blas_closest_packet(sc, it, lr, in, gbound, s_stack, 0, &tb)) {
        if (in) blas_closest<true>(sc, it, lr, gbound, s_stack, 0, &tb);
    }
    if (in && tb.found) {
        const float t = tb.t;
        const uint32_t face = tb.slot | (tb.side << 30); // bit31 back face, bit30 negated normal
        const bool better = !best->found || t < best->t || (t == best->t && (key < best->key || (key == best->key && idx < best->item)));
        if (better) { best->found = true; best->t = t; best->item = idx; best->face = face; best->key = key; }
    }
}
#else
RR_DEV void closest_item_packet(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, int* s_stack, Closest* best) { closest_item(sc, idx, o, d, depth, s_stack, 0, best); }
#endif

RR_DEV void trace_closest_ray(const DSceneView& sc, f3 o, f3 d, uint32_t depth, int* s_stack, Closest* best) {
    best->found = false; best->nan_seen = false; best->t = RR_FLT_MAX; best->item = -1; best->face = 0u; best->key = 0.0f;
    // top level: world-space boxes over items (stands in for Scene::get_possible_hits_by_ray,
    // reference src/scene.rs:1715-1722; any conservative candidate set gives the same result)
    // the top level in the 4-wide form of the per-mesh trees, same step (sentinel-terminated stack)
    const Slab4 ws = make_slab4(make_slab(o, d), 0u);
    int sp = 1;
    STK(0) = RR_SENTINEL;
    int cur = sc.tlas_root4c; // (the closest-hit tree: surface boxes)
    // while-while: every lane walks the top level until it holds a candidate item (or is done), so the per-mesh
    // walks below run with the lanes of the wave together instead of one straggler at a time
    for (;;) {
        while (cur >= 0) { RR_NODE4_STEP(sc.tnodes4c, ws, fminf(best->t * RR_TOI_SLACK, RR_FLT_MAX)) }
        if (cur == RR_SENTINEL) break;
        closest_item(sc, (int)RR_LEAF_FIRST((uint32_t)~cur), o, d, depth, s_stack, sp, best); // one item per top-level leaf
        sp--; cur = STK(sp);
    }
}

// Rays with a non-finite component (a NaN normal, e.g. from a normal map on a sphere whose tangent degenerates, reflects
// into one).  The reference has no special case for them and its arithmetic decides: in item-local space such a ray
// is NaN in all components of its origin or direction, ray_toi_with_ball's comparisons are then all false and EVERY
// candidate sphere reports Some(NaN); a triangle's toi comes out NaN or infinite and fails `toi <= max_toi`.  The
// candidate loop (src/raytracing.rs:466-487) keeps the first such sphere in (bbox distance, item) order, since nothing
// compares smaller than NaN, and the hit shades with NaN position and normal (texel (0, 0), finite ambient term).
// The top-level walk has no defined order for these rays (NaN passes or fails a slab test by the instruction used), so they take this walk over the items
// instead: exact for spheres; meshes are skipped, which is what the reference's triangle test amounts to.
RR_DEV bool ray_nonfinite(f3 o, f3 d) {
    const float z = ((o.x - o.x) + (o.y - o.y) + (o.z - o.z)) + ((d.x - d.x) + (d.y - d.y) + (d.z - d.z)); // x - x: 0 for finite x, NaN otherwise
    return z != 0.0f;
}
RR_DEV void trace_closest_nonfinite(const DSceneView& sc, f3 o, f3 d, uint32_t depth, Closest* best) {
    best->found = false; best->nan_seen = false; best->t = RR_FLT_MAX; best->item = -1; best->face = 0u; best->key = 0.0f;
    Closest first = *best; // the first candidate in the reference's order that is hit at all
    for (int idx = 0; idx < (int)sc.n_items; idx++) {
        const DItem& it = rr_global(sc.items)[idx];
        const uint32_t flags = it.flags;
        if (!(flags & RR_IF_SPHERE) || !item_passes(flags, false, depth)) continue;
        LRay lr = inverse_ray(it, o, d, true); // (w is NaN for a non-finite origin: see to_local_point)
        float key, t; bool inside;
        if (!aabb_cast(it.bmin, it.bmax, lr, (flags & RR_IF_SOLID_BASE) != 0u, &key) || key != key) continue;
        if (!ray_ball(it.radius, lr, (flags & RR_IF_SOLID_BASE) != 0u, &t, &inside)) continue;
        if (!first.found || key < first.key || (key == first.key && idx < first.item)) { first.found = true; first.t = t; first.item = idx; first.key = key; }
        if (t == t && (!best->found || t < best->t || (t == best->t && (key < best->key || (key == best->key && idx < best->item))))) {
            best->found = true; best->t = t; best->item = idx; best->key = key;
        }
    }
    if (first.found && first.t != first.t) *best = first; // a NaN toi is never replaced (`toi < best` is false)
}

// A FINITE ray can get Some(NaN) from a ball too: where ray_toi_with_ball's products overflow (a ball of radius 1e12 under a
// transform that shrinks it to one unit: b * b = inf, a * c = inf, delta = NaN, every comparison false).  The reference's loop
// (src/raytracing.rs:466-487) walks the candidates in (bbox distance, item) order and replaces its best hit on `toi < best`: a NaN
// hit is THE result if it is the first candidate hit at all in that order (nothing compares smaller than NaN afterwards) and is
// ignored otherwise.  The walks above visit candidates in another order and keep a minimum, which is only order-free while every
// toi is a number; they leave a NaN hit out and flag the ray (Closest::nan_seen), and the flagged rays -- none in any scene whose
// balls have sane sizes -- take this pass over all items in the reference's own terms.
RR_DEV void trace_closest_ordered(const DSceneView& sc, f3 o, f3 d, uint32_t depth, int* s_stack, Closest* best) {
    Closest fin; fin.found = false; fin.nan_seen = false; fin.t = RR_FLT_MAX; fin.item = -1; fin.face = 0u; fin.key = 0.0f;
    Closest first = fin; // the first candidate in the reference's order that is hit at all
    for (int idx = 0; idx < (int)sc.n_items; idx++) {
        const DItem& it = rr_global(sc.items)[idx];
        const uint32_t flags = it.flags;
        if (!item_passes(flags, false, depth)) continue;
        const LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
        const bool solid = (flags & RR_IF_SOLID_BASE) != 0u;
        float key, t; uint32_t face = 0u;
        if (!aabb_cast(it.bmin, it.bmax, lr, solid, &key) || key != key) continue;
        if (flags & RR_IF_SPHERE) {
            bool inside;
            if (!ray_ball(it.radius, lr, solid, &t, &inside)) continue;
        } else {
            if (it.n_tris == 0u) continue;
            TriBest tb;
            blas_closest<true>(sc, it, lr, RR_FLT_MAX, s_stack, 0, &tb);
            if (!tb.found) continue;
            t = tb.t; face = tb.slot | (tb.side << 30);
        }
        if (!first.found || key < first.key || (key == first.key && idx < first.item)) { first.found = true; first.t = t; first.item = idx; first.face = face; first.key = key; }
        if (t == t && (!fin.found || t < fin.t || (t == fin.t && (key < fin.key || (key == fin.key && idx < fin.item))))) {
            fin.found = true; fin.t = t; fin.item = idx; fin.face = face; fin.key = key;
        }
    }
    *best = (first.found && first.t != first.t) ? first : fin;
}

// ---------------------------------------------------------------------------
// The top level for a coherent packet.  The top level is only a candidate filter: every item it lets through is
// tested exactly in its own space, and the winner is a minimum that does not depend on the order.  A packet whose 64
// rays share their direction signs (64 samples of one pixel do) therefore does not walk the top-level tree 64 times:
// the wave bounds its rays by an interval ray (component ranges of origin and reciprocal direction), tests the items'
// world boxes against it with one ITEM per lane, and all lanes then visit the few candidates together, nearest box
// first, until the next box starts behind every lane's best hit.  On the contract frame the per-ray walk spent a third
// of the kernel's vector instructions in the top level (7.8 node steps per ray over 194 items).
// All 64 lanes must be active.  Returns false (nothing touched) when the packet is not coherent, the scene has more
// items than a few passes cover, or more than 64 items survive: the caller walks the tree per ray instead.
// ---------------------------------------------------------------------------
RR_DEV float wave_min_f32(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));  // quad_perm [1,0,3,2]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));  // quad_perm [2,3,0,1]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false))); // row_half_mirror
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false))); // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
RR_DEV float wave_max_f32(float v) { return -wave_min_f32(-v); }
RR_DEV uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}
// one axis of the interval-ray slab test: lower bound of the entry distance and upper bound of the exit distance over
// all rays with origin in [olo, ohi] and |1/d| in [alo, ahi], direction sign `neg` (wave-uniform)
RR_DEV void beam_axis(bool neg, float blo, float bhi, float olo, float ohi, float alo, float ahi, float* tn, float* tf) {
    const float un = neg ? olo - bhi : blo - ohi; // smallest signed distance to the near plane
    const float wf = neg ? ohi - blo : bhi - olo; // largest signed distance to the far plane
    *tn = un * (un >= 0.0f ? alo : ahi);
    *tf = wf * (wf >= 0.0f ? ahi : alo);
}
// The candidate list of a packet: lane l of the wave holds candidate l as (sort key, item); the key keeps the upper bits
// of the distance at which the item's box can first be entered by any ray of the packet (a lower bound) and the lane in
// its low six bits, 0xffffffff = none.  `far`: (wave-uniform) boxes that start beyond it are of no interest.
// `boxes`: sc.item_boxes (the items' corner boxes: shadow packets, whose order and bounds are the LOCAL boxes' entry distances, which only a world box
// around the local box bounds from below) or sc.item_boxes + 2 * n_items (their surface boxes: closest-hit packets; rr_api.hip build_tlas).
// (A shadow packet that also dropped the items whose surface box none of its rays reaches gained nothing: 6.00 -> 6.02 ms.)
RR_DEV bool beam_candidates(const DSceneView& sc, const float4* __restrict__ boxes, f3 o, f3 d, float far, int* s_stack, uint32_t* sk_out, int* item_out, uint32_t min_items = RR_BEAM_MIN_ITEMS) {
    const uint32_t n_items = sc.n_items;
    if (n_items > RR_BEAM_MAX_ITEMS || n_items < min_items) return false;
    // coherent: finite rays, no zero direction component, one sign per axis
    const bool bad = ray_nonfinite(o, d) || !(fabsf(d.x) > 1e-30f) || !(fabsf(d.y) > 1e-30f) || !(fabsf(d.z) > 1e-30f);
    const unsigned long long nx_ = __ballot(d.x < 0.0f), ny_ = __ballot(d.y < 0.0f), nz_ = __ballot(d.z < 0.0f);
    if (__ballot(bad) != 0ull || (nx_ != 0ull && ~nx_ != 0ull) || (ny_ != 0ull && ~ny_ != 0ull) || (nz_ != 0ull && ~nz_ != 0ull)) return false;
    const bool negx = nx_ != 0ull, negy = ny_ != 0ull, negz = nz_ != 0ull;
    const float ax = fabsf(__builtin_amdgcn_rcpf(d.x)), ay = fabsf(__builtin_amdgcn_rcpf(d.y)), az = fabsf(__builtin_amdgcn_rcpf(d.z));
    const float oxl = wave_min_f32(o.x), oxh = wave_max_f32(o.x), oyl = wave_min_f32(o.y), oyh = wave_max_f32(o.y), ozl = wave_min_f32(o.z), ozh = wave_max_f32(o.z);
    const float axl = wave_min_f32(ax) * 0.99999f, axh = wave_max_f32(ax) * 1.00001f;
    const float ayl = wave_min_f32(ay) * 0.99999f, ayh = wave_max_f32(ay) * 1.00001f;
    const float azl = wave_min_f32(az) * 0.99999f, azh = wave_max_f32(az) * 1.00001f;
    const uint32_t lane = threadIdx.x & (RR_WAVE - 1), wave_col = threadIdx.x & ~(RR_WAVE - 1u);
    // the items' boxes against the interval ray, one item per lane; survivors appended to a list in the wave's own
    // columns of two stack rows (nothing is on the stack yet)
    uint32_t total = 0;
    for (uint32_t base = 0; base < n_items; base += RR_WAVE) {
        const uint32_t j = base + lane;
        bool cand = false; float key = 0.0f;
        if (j < n_items) {
            const float4 lo = boxes[2u * j], hi = boxes[2u * j + 1u];
            float tnx, tfx, tny, tfy, tnz, tfz;
            beam_axis(negx, lo.x, hi.x, oxl, oxh, axl, axh, &tnx, &tfx);
            beam_axis(negy, lo.y, hi.y, oyl, oyh, ayl, ayh, &tny, &tfy);
            beam_axis(negz, lo.z, hi.z, ozl, ozh, azl, azh, &tnz, &tfz);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
            const float tf = fminf(fminf(tfx, tfy), tfz);
            key = tn * (1.0f / RR_TOI_SLACK) * 0.99999f; // a lower bound on any toi the item can report
            cand = key <= tf * 1.00001f && key <= far;
        }
        const unsigned long long m = __ballot(cand);
        const uint32_t pos = total + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        total += (uint32_t)__popcll(m);
        if (total > RR_WAVE) return false; // (wave-uniform) more candidates than lanes: not a packet worth treating as one
        if (cand) { s_stack[1 * RR_BLOCK + wave_col + pos] = __float_as_int(key); s_stack[2 * RR_BLOCK + wave_col + pos] = (int)j; }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t sk = 0xffffffffu; int item = 0;
    if (lane < total) { sk = ((uint32_t)s_stack[1 * RR_BLOCK + wave_col + lane] & ~63u) | lane; item = s_stack[2 * RR_BLOCK + wave_col + lane]; }
    __builtin_amdgcn_wave_barrier();
    *sk_out = sk; *item_out = item;
    return true;
}
// next candidate in box-distance order: (wave-uniform) false when none is left; *key = the lower bound of its box distance
RR_DEV bool beam_next(uint32_t& sk, int item, float* key, int* idx) {
    const uint32_t m = wave_min_u32(sk);
    if (m == 0xffffffffu) return false;
    const uint32_t src = m & 63u;
    *key = __uint_as_float(m & ~63u);
    *idx = __builtin_amdgcn_readlane(item, src);
    if ((threadIdx.x & (RR_WAVE - 1)) == src) sk = 0xffffffffu;
    return true;
}
RR_DEV bool trace_closest_packet(const DSceneView& sc, f3 o, f3 d, uint32_t depth, int* s_stack, Closest* best) {
    uint32_t sk; int item;
    if (!beam_candidates(sc, sc.item_boxes + 2u * sc.n_items, o, d, RR_FLT_MAX, s_stack, &sk, &item, RR_BEAM_MIN_ITEMS_CLOSEST)) return false;
    best->found = false; best->nan_seen = false; best->t = RR_FLT_MAX; best->item = -1; best->face = 0u; best->key = 0.0f;
    float key; int idx;
    while (beam_next(sk, item, &key, &idx)) {
        // the remaining boxes all start at or behind this one: done when that is behind every lane's best hit
        if (__ballot(!best->found || key <= best->t) == 0ull) break;
        closest_item_packet(sc, idx, o, d, depth, s_stack, best);
    }
    return true;
}

// Shadow rays stop at the first ITEM (in bbox-distance order) that is hit at all
// (reference src/raytracing.rs:483-486), not at the nearest hit.
struct ShadowSel { float key; int item; bool found; bool within; float t; uint32_t face; };

RR_DEV void shadow_item(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, float limit,
                        int* s_stack, int sp_base, ShadowSel* sel) {
    RR_UTIL(1)
    const DItem& it = rr_global(sc.items)[idx];
    uint32_t flags = it.flags;
    if (!item_passes(flags, true, depth)) return;
    LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
    float key, tmin;
    if (!aabb_cast2(it.bmin, it.bmax, lr, false, &key, &tmin)) return; // for_shadow forces solid = false
    if (key != key) return;
    // An item whose box starts beyond the light can never be hit within the light distance; it is skipped in this
    // pass.  It can still matter as a BLOCKER (hit, ordered before the occluder found here): trace_shadow_ray
    // runs a second pass for exactly that case.
    if (tmin > limit * RR_TOI_SLACK) return;
    if (sel->found && !(key < sel->key || (key == sel->key && idx < sel->item))) return;
    bool any = false, within = false; float t = 0.0f; uint32_t face = 0u;
    if (flags & RR_IF_SPHERE) {
        bool inside;
        if (ray_ball(it.radius, lr, false, &t, &inside)) { any = true; within = !(t > limit); } // (`in_light = toi > len`, :890: false for a NaN toi)
    } else if (it.n_tris != 0u) {
        if (flags & RR_IF_OCCLUDER_ALPHA_TEX) { // the occluder's alpha map needs the true nearest hit
            TriBest tb;
            blas_closest<false>(sc, it, lr, RR_FLT_MAX, s_stack, sp_base, &tb);
            if (tb.found) { any = true; t = tb.t; within = t <= limit; face = tb.face + ((tb.side & 2u) ? it.n_tris : 0u); }
        } else {
            blas_any(sc, it, lr, limit, s_stack, sp_base, &any, &within);
        }
    }
    if (any) { sel->found = true; sel->key = key; sel->item = idx; sel->within = within; sel->t = t; sel->face = face; }
}

// shadow_item for a candidate that all lanes of a packet visit together (trace_shadow_packet): a mesh without an alpha map is
// walked once per wave (blas_any_packet).  Same result as shadow_item, lane by lane.  OFF by default (-DRR_PACKET_WALK_SHADOW):
// parity-green, but any-hit lanes leave their walk one by one as they find a hit and a shared walk goes on for the rest -- level-1
// shadow time 5.96 -> 6.02 ms on sponza_syn, where the closest-hit kernel gains 3 % from the same form.
#ifdef RR_PACKET_WALK_SHADOW
RR_DEV void shadow_item_packet(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, float limit, int* s_stack, ShadowSel* sel) {
    const DItem& it = rr_global(sc.items)[idx];
    const uint32_t flags = it.flags;
    if (flags & (RR_IF_SPHERE | RR_IF_OCCLUDER_ALPHA_TEX)) { shadow_item(sc, idx, o, d, depth, limit, s_stack, 0, sel); return; } // (wave-uniform)
    RR_UTIL(1)
    bool in = item_passes(flags, true, depth) && it.n_tris != 0u;
    const LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
    float key = 0.0f, tmin = 0.0f;
    in = in && aabb_cast2(it.bmin, it.bmax, lr, false, &key, &tmin); // for_shadow forces solid = false
    in = in && key == key && !(tmin > limit * RR_TOI_SLACK);
    in = in && !(sel->found && !(key < sel->key || (key == sel->key && idx < sel->item)));
    if (__ballot(in) == 0ull) return;
    bool any = false, within = false;
    if (!blas_any_packet(sc, it, lr, in, limit, s_stack, 0, &any, &within)) {
        if (in) blas_any(sc, it, lr, limit, s_stack, 0, &any, &within);
    }
    if (in && any) { sel->found = true; sel->key = key; sel->item = idx; sel->within = within; sel->t = 0.0f; sel->face = 0u; }
}
#else
RR_DEV void shadow_item_packet(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, float limit, int* s_stack, ShadowSel* sel) { shadow_item(sc, idx, o, d, depth, limit, s_stack, 0, sel); }
#endif

// Second pass of a shadow query: is there an item whose box starts BEYOND the light (skipped above), ordered before
// the selected occluder (sel), that is hit at all?  The reference tries candidates in bbox-distance order and the
// first one that is hit decides (src/raytracing.rs:466-487); its hit lies beyond the light, so the receiver is lit --
// unless it is a ball whose arithmetic overflowed into Some(NaN): `in_light = toi > len` is false for that, the receiver is
// in ITS shadow (trace_closest_ordered has the story).  Returns 0: no blocker, 1: a hit beyond the light, 2: a NaN hit.
RR_DEV int shadow_blocker_item(const DSceneView& sc, int idx, f3 o, f3 d, uint32_t depth, float limit, const ShadowSel& sel,
                               int* s_stack, int sp_base, float* key_out) {
    const DItem& it = rr_global(sc.items)[idx];
    const uint32_t flags = it.flags;
    if (!item_passes(flags, true, depth)) return 0;
    LRay lr = inverse_ray(it, o, d, sc.general_w != 0u);
    float key, tmin;
    if (!aabb_cast2(it.bmin, it.bmax, lr, false, &key, &tmin)) return 0;
    if (key != key || !(tmin > limit * RR_TOI_SLACK)) return 0;
    if (!(key < sel.key || (key == sel.key && idx < sel.item))) return 0;
    *key_out = key;
    if (flags & RR_IF_SPHERE) { float t; bool inside; return ray_ball(it.radius, lr, false, &t, &inside) ? (t != t ? 2 : 1) : 0; }
    if (it.n_tris == 0u) return 0;
    bool any = false, within = false;
    blas_any(sc, it, lr, RR_FLT_MAX, s_stack, sp_base, &any, &within);
    return any ? 1 : 0;
}

// Updates *sel to the outcome: lit (within = false) if the first blocker in the reference's order has a hit beyond the light, in the
// shadow of that blocker if its toi is NaN, untouched without a blocker.  A scene without balls that can overflow (RR_VIEW_NAN_BALLS,
// the usual case) is done at the first blocker found: they all say "lit".
RR_DEV void trace_shadow_blockers(const DSceneView& sc, f3 o, f3 d, uint32_t depth, float limit, ShadowSel* sel, int* s_stack) {
    const float bound = sel->key * 1.00001f + 1e-6f; // a blocker's box starts before the occluder's key
    const bool nan_balls = (sc.compat & RR_VIEW_NAN_BALLS) != 0u;
    ShadowSel first = *sel; // the first blocker in (key, item) order so far; starts as the occluder it must precede
    int first_kind = 0;
    const Slab4 ws = make_slab4(make_slab(o, d), 0u);
    int sp = 1;
    STK(0) = RR_SENTINEL;
    int cur = sc.tlas_root4;
    for (;;) {
        while (cur >= 0) { RR_NODE4_STEP_PLAIN(sc.tnodes4, ws, bound) }
        if (cur == RR_SENTINEL) break;
        const int idx = (int)RR_LEAF_FIRST((uint32_t)~cur);
        float key = 0.0f;
        const int kind = shadow_blocker_item(sc, idx, o, d, depth, limit, first, s_stack, sp, &key);
        if (kind != 0) {
            first.key = key; first.item = idx; first_kind = kind;
            if (!nan_balls) break;
        }
        sp--; cur = STK(sp);
    }
    if (first_kind == 1) sel->within = false;
    else if (first_kind == 2) { sel->key = first.key; sel->item = first.item; sel->within = true; sel->t = __uint_as_float(0x7fc00000u); sel->face = 0u; }
}

// A non-finite shadow ray (a NaN normal puts the origin at NaN) in the reference: every candidate sphere reports Some(NaN)
// (see trace_closest_nonfinite), triangles report nothing, the first sphere in (bbox distance, item) order is THE
// intersection, and `in_light = toi > len` is false for a NaN toi -- the receiver is in shadow, for every kind of light
// (src/raytracing.rs:884-892).  Its alpha map, if it has one, is then sampled at a NaN uv (a NaN texel under the
// bilinear filter: the sample turns the pixel white).  Found by tools/fuzz_parity.py rich, seed 6601.
RR_DEV void trace_shadow_nonfinite(const DSceneView& sc, f3 o, f3 d, uint32_t depth, float limit, ShadowSel* sel) {
    sel->found = false; sel->within = false; sel->key = 0.0f; sel->item = -1; sel->t = 0.0f; sel->face = 0u;
    for (int idx = 0; idx < (int)sc.n_items; idx++) {
        const DItem& it = sc.items[idx];
        const uint32_t flags = it.flags;
        if (!(flags & RR_IF_SPHERE) || !item_passes(flags, true, depth)) continue;
        LRay lr = inverse_ray(it, o, d, true); // (w is NaN for a non-finite origin: see to_local_point)
        float key, tmin, t; bool inside;
        if (!aabb_cast2(it.bmin, it.bmax, lr, false, &key, &tmin) || key != key) continue; // for_shadow forces solid = false
        if (!ray_ball(it.radius, lr, false, &t, &inside)) continue;
        if (!sel->found || key < sel->key || (key == sel->key && idx < sel->item)) {
            sel->found = true; sel->key = key; sel->item = idx; sel->t = t; sel->within = !(t > limit);
        }
    }
}

RR_DEV void trace_shadow_ray(const DSceneView& sc, f3 o, f3 d, uint32_t depth, float limit, int* s_stack, ShadowSel* sel) {
    if (ray_nonfinite(o, d)) { trace_shadow_nonfinite(sc, o, d, depth, limit, sel); return; }
    sel->found = false; sel->within = false; sel->key = 0.0f; sel->item = -1; sel->t = 0.0f; sel->face = 0u;
    // an item whose world box starts beyond the light, or beyond the selected item's key, cannot matter
#define RR_SHADOW_BOUND fminf(sel->found ? fminf(limit * RR_TOI_SLACK, sel->key * 1.00001f + 1e-6f) : limit * RR_TOI_SLACK, RR_FLT_MAX)
    const Slab4 ws = make_slab4(make_slab(o, d), 0u);
    int sp = 1;
    STK(0) = RR_SENTINEL;
    int cur = sc.tlas_root4;
    for (;;) {
        while (cur >= 0) { RR_NODE4_STEP_PLAIN(sc.tnodes4, ws, RR_SHADOW_BOUND) }
        if (cur == RR_SENTINEL) break;
        shadow_item(sc, (int)RR_LEAF_FIRST((uint32_t)~cur), o, d, depth, limit, s_stack, sp, sel);
        sp--; cur = STK(sp);
    }
    // The occluder found has a hit within the light distance.  Only if its sort key lies beyond the light (its box
    // contains the ray origin, so the key is the box EXIT distance) can an item that starts beyond the light precede it.
    if (sel->found && sel->within && sel->key > limit) trace_shadow_blockers(sc, o, d, depth, limit, sel, s_stack);
}

// The packet form of trace_shadow_ray's first pass (see trace_closest_packet): candidates in box-distance order, until the
// next box starts beyond every lane's bound (the light, or the key of the occluder selected so far).
RR_DEV bool trace_shadow_packet(const DSceneView& sc, f3 o, f3 d, uint32_t depth, float limit, int* s_stack, ShadowSel* sel) {
    uint32_t sk; int item;
    if (!beam_candidates(sc, sc.item_boxes, o, d, wave_max_f32(limit == limit ? limit : 0.0f), s_stack, &sk, &item)) return false;
    sel->found = false; sel->within = false; sel->key = 0.0f; sel->item = -1; sel->t = 0.0f; sel->face = 0u;
    float key; int idx;
    while (beam_next(sk, item, &key, &idx)) {
        if (__ballot(key <= RR_SHADOW_BOUND) == 0ull) break; // (a NaN light distance compares false: that lane wants nothing, as in the per-ray walk)
        shadow_item_packet(sc, idx, o, d, depth, limit, s_stack, sel);
    }
    if (sel->found && sel->within && sel->key > limit) trace_shadow_blockers(sc, o, d, depth, limit, sel, s_stack);
    return true;
}

// ---------------------------------------------------------------------------
// textures: reference src/raytracing.rs:629-675, src/shape/mod.rs:510-629
// ---------------------------------------------------------------------------
// `lut`: the u8 -> f32 table (c_u8_to_f32, or a workgroup's copy of it in LDS: four dependent reads per texel)
RR_DEV float4 texel(const DSceneView& sc, const DTexture& t, uint32_t x, uint32_t y, const float* lut = c_u8_to_f32) {
    uint32_t p = rr_global(sc.texels)[t.offset + (uint64_t)y * t.width + x];
    return make_float4(lut[p & 255u], lut[(p >> 8) & 255u], lut[(p >> 16) & 255u], lut[p >> 24]);
}
RR_DEV uint32_t tex_wrap(float val, uint32_t bound) {
    int32_t sb = (int32_t)bound;
    const int32_t x = as_i32(val * (float)bound);
    // power-of-two sizes: the mask IS the remainder made non-negative (x % sb, plus sb when negative), without the integer division
    if ((bound & (bound - 1u)) == 0u) return (uint32_t)x & (bound - 1u);
    int32_t w = x % sb;
    return (w < 0) ? (uint32_t)(w + sb) : (uint32_t)w;
}
RR_DEV float lerp1(float a, float b, float f) { return a + f * (b - a); } // helper::interpolate
RR_DEV float4 tex_bilinear(const DSceneView& sc, const DTexture& t, float u, float v, const float* lut = c_u8_to_f32) {
    uint32_t width = t.width, height = t.height;
    float x = u * (float)width, y = v * (float)height;
    if (x < 0.0f) x = x + (float)width;
    if (y < 0.0f) y = y + (float)height;
    uint32_t x0 = as_u32(floorf(x)), x1 = as_u32(ceilf(x));
    uint32_t y0 = as_u32(floorf(y)), y1 = as_u32(ceilf(y));
    if (x0 >= width) x0 = width - 1u;
    if (y0 >= height) y0 = height - 1u;
    if (x1 >= width) x1 = width - 1u;
    if (y1 >= height) y1 = height - 1u;
    float fx = x - (float)x0, fy = y - (float)y0;
    float4 p0 = texel(sc, t, x0, y0, lut), p1 = texel(sc, t, x1, y0, lut), p2 = texel(sc, t, x0, y1, lut), p3 = texel(sc, t, x1, y1, lut);
    float4 a = make_float4(lerp1(p0.x, p1.x, fx), lerp1(p0.y, p1.y, fx), lerp1(p0.z, p1.z, fx), lerp1(p0.w, p1.w, fx));
    float4 b = make_float4(lerp1(p2.x, p3.x, fx), lerp1(p2.y, p3.y, fx), lerp1(p2.z, p3.z, fx), lerp1(p2.w, p3.w, fx));
    return make_float4(lerp1(a.x, b.x, fy), lerp1(a.y, b.y, fy), lerp1(a.z, b.z, fy), lerp1(a.w, b.w, fy));
}
// The material of a hit, copied into registers once (four 16-B loads + the flag word): the shading code stores to
// queues and accumulators between its uses, so fields read through the pointer would be re-fetched one dword at a
// time, each fetch a dependent round trip.  Texture slots are only looked at when the flag word says they are set.
struct MatR {
    f3 ambient, base, specular;
    float alpha, shininess, reflectivity, refraction_index, normal_map_strength, shadow_softness, roughness;
    float cos_shadow_softness, cos_roughness; // jitter()'s z_lo for the two constant spreads (host-evaluated, DMaterial)
    uint32_t flags;
    const DMaterial* p;
    const float* lut; // u8 -> f32 table of the workgroup (LDS)
};
RR_DEV MatR load_material(const DMaterial* p, const float* lut) {
    const float4* q = (const float4*)p;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[6]; // q[4], q[5]: the texture slots, read when a slot's flag is set
    MatR m;
    m.ambient = mk3(a.x, a.y, a.z); m.alpha = a.w;
    m.base = mk3(b.x, b.y, b.z); m.shininess = b.w;
    m.specular = mk3(c.x, c.y, c.z); m.reflectivity = c.w;
    m.refraction_index = d.x; m.normal_map_strength = d.y; m.shadow_softness = d.z; m.roughness = d.w;
    m.flags = __float_as_uint(e.x); m.cos_shadow_softness = e.y; m.cos_roughness = e.z; m.p = p; m.lut = lut;
    return m;
}
RR_DEV bool tex_color(const DSceneView& sc, const MatR& m, bool has_uv, f2 uv, int slot, float4* out) {
    if (!(m.flags & (RR_MF_TEX_SLOT0 << slot)) || !has_uv) return false; // slot bit = index >= 0 and width > 0
    DTexture t; // the slot's descriptor sits in the material record itself (DMaterial::texd): one 16-B load at an address known since the material was
    { const uint4 q = *(const uint4*)&m.p->texd[slot]; t.offset = (uint64_t)q.x | ((uint64_t)q.y << 32); t.width = q.z; t.height = q.w; }
    if (m.flags & RR_MF_NEAREST) *out = texel(sc, t, tex_wrap(uv.x, t.width), tex_wrap(uv.y, t.height), m.lut);
    else *out = tex_bilinear(sc, t, uv.x, uv.y, m.lut);
    return true;
}
// get_tex_color: false = None
RR_DEV bool tex_color(const DSceneView& sc, const DMaterial& m, bool has_uv, f2 uv, int slot, float4* out) {
    int ti = m.tex[slot];
    if (ti < 0 || !has_uv) return false;
    DTexture t; t.offset = m.texd[slot].offset; t.width = m.texd[slot].width; t.height = m.texd[slot].height;
    if (t.width == 0u) return false;
    if (m.flags & RR_MF_NEAREST) *out = texel(sc, t, tex_wrap(uv.x, t.width), tex_wrap(uv.y, t.height));
    else *out = tex_bilinear(sc, t, uv.x, uv.y);
    return true;
}

// ---------------------------------------------------------------------------
// uv / normals: reference src/shape/mesh.rs:105-161, :204-259; src/shape/sphere.rs:69-99
// ---------------------------------------------------------------------------
RR_DEV void area_weights(f3 a, f3 b, f3 c, f3 p, float* a1, float* a2, float* a3) {
    f3 f1 = a - p, f2v = b - p, f3v = c - p;
    float area = norm3(cross3(a - b, a - c));
    *a1 = norm3(cross3(f2v, f3v)) / area;
    *a2 = norm3(cross3(f3v, f1)) / area;
    *a3 = norm3(cross3(f1, f2v)) / area;
}
// the same with the triangle's area from the host (DTri::v1.w: the value of the line `area = ...` above, bit for bit)
RR_DEV void area_weights(f3 a, f3 b, f3 c, f3 p, float area, float* a1, float* a2, float* a3) {
    f3 f1 = a - p, f2v = b - p, f3v = c - p;
    *a1 = norm3(cross3(f2v, f3v)) / area;
    *a2 = norm3(cross3(f3v, f1)) / area;
    *a3 = norm3(cross3(f1, f2v)) / area;
}
RR_DEV f2 sphere_uv(const DItem& it, f3 hit, bool general_w) {
    f3 p = to_local_point(it, hit, general_w);
    float theta = rr_atan2(-(p.z - 0.0f), p.x - 0.0f);
    float u = (theta + RR_PI_F) / (2.0f * RR_PI_F);
    float phi = rr_acos((-(p.y - 0.0f)) / it.radius);
    float v = phi / RR_PI_F;
    f2 r; r.x = u; r.y = -v; return r;
}
RR_DEV f2 mesh_uv(const DSceneView& sc, const DItem& it, uint32_t slot, f3 hit, bool general_w) {
    f2 r; r.x = 0.0f; r.y = 0.0f;
    const DTriAttr at = rr_global(sc.attrs)[it.tri_base + slot];
    if (!(__float_as_uint(at.s3.w) & 1u)) return r;
    f3 p = to_local_point(it, hit, general_w);
    const DTri tr = rr_global(sc.tris)[it.tri_base + slot];
    float a1, a2, a3;
    area_weights(mk3(tr.v0.x, tr.v0.y, tr.v0.z), mk3(tr.v1.x, tr.v1.y, tr.v1.z), mk3(tr.v2.x, tr.v2.y, tr.v2.z), p, &a1, &a2, &a3);
    float ux = (at.s0.w * a1 + at.s2.w * a2) + at.s3.y * a3;
    float uy = (at.s1.w * a1 + at.s3.x * a2) + at.s3.z * a3;
    r.x = ux; r.y = -uy;
    return r;
}

// ---------------------------------------------------------------------------
// jitter (reference src/raytracing.rs:565-626) on the counter-based generator
// ---------------------------------------------------------------------------
struct RngKey { uint32_t seed_lo, seed_hi, pixel, sample, node; };
// z_lo = rr_cos(spread * RR_PI_F): passed in, because for the two spreads that are material constants (shadow_softness, roughness
// without a map) the host has evaluated it once per material (DMaterial::cos_*) with the same rr_cos
RR_DEV f3 jitter(f3 dir, float spread, float z_lo, const RngKey& k, uint32_t stream) {
    if (spread <= 0.0f) return dir;
    f3 b3 = normalize3(dir);
    f3 diff = (rr_abs(b3.x) < 0.5f) ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 1.0f, 0.0f);
    f3 b1 = normalize3(cross3(b3, diff));
    f3 b2 = cross3(b1, b3);
    if (!(z_lo < 1.0f)) return dir;
    uint32_t r0, r1;
    philox4x32_10(k.pixel, k.sample, k.node, stream, k.seed_lo, k.seed_hi, &r0, &r1);
    float z = uniform_f32(r0, z_lo, 1.0f);
    float r = sqrtf(1.0f - z * z);
    float theta = uniform_f32(r1, -RR_PI_F, RR_PI_F);
    float s, c;
    rr_sincos(theta, &s, &c);
    float x = r * c, y = r * s;
    f3 nd = (x * b1 + y * b2) + z * b3;
    return normalize3(nd);
}

// fresnel, reference src/raytracing.rs:535-563 (cos_i = |cos_t| as written there)
RR_DEV float fresnel(f3 incident, f3 normal, float index) {
    float i_dot_n = dot3(incident, normal);
    float eta_i = 1.0f, eta_t = index;
    if (i_dot_n > 0.0f) { eta_i = eta_t; eta_t = 1.0f; }
    float sin_t = eta_i / eta_t * sqrtf(rs_max(1.0f - i_dot_n * i_dot_n, 0.0f));
    if (sin_t > 1.0f) return 1.0f;
    float cos_t = sqrtf(rs_max(1.0f - sin_t * sin_t, 0.0f));
    float cos_i = rr_abs(cos_t);
    float r_s = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    float r_p = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_s * r_s + r_p * r_p) / 2.0f;
}

// ---------------------------------------------------------------------------
// accumulators
// ---------------------------------------------------------------------------
RR_DEV long long to_fix(float v, float scale, float clampv) {
    if (v != v) return 0ll; // a NaN adds nothing to the fixed-point sum; its pixel is flagged (to_fix_c)
    v = fminf(fmaxf(v, -clampv), clampv);
    return __float2ll_rn(v * scale);
}
// The fixed-point sum of one lane and channel is a 32-bit integer.  Nearly every term is small (a colour term up to 2, a normal
// component, a depth below 512: |v * scale| < 2^25) and converts with v_rndne + v_cvt_i32; __float2ll_rn is eleven instructions and
// a 64-bit add two more, seven to ten times per shaded hit.  A term beyond 2^25 goes straight to its accumulator word as one 64-bit
// atomic of its own (rare: a saturating highlight, a far hit's depth); a NaN adds nothing (its pixel is flagged).  Every term is
// rounded to nearest-even by itself either way, and integer adds commute: the pixel's sum is the same integer.
RR_DEV void fix_add(int& n, float v, float scale, float clampv, long long* plane, uint32_t pix) {
    const float x = v * scale;
    if (rr_abs(x) < 33554432.0f) n += __float2int_rn(x); // (inside every clamp in use: |v| < 2 at scale 2^24, < 512 at 2^16)
    else { const long long t = to_fix(v, scale, clampv); if (t != 0ll && plane) atomicAdd((unsigned long long*)plane + pix, (unsigned long long)t); }
}

// A colour contribution of channel `ch`: non-finite values are recorded in `flags` (RR_NF_*), because the reference's f32
// sum would carry them to the pixel (NaN or +inf -> 255, -inf -> 0, src/raytracing.rs:406-417) while a fixed-point sum cannot.
RR_DEV uint32_t nonfinite_flags(float r, float g, float b) {
    if (((r - r) + (g - g)) + (b - b) == 0.0f) return 0u; // x - x is 0 for every finite x and NaN otherwise: one test for the common case
    uint32_t f = 0u;
    const float v[3] = {r, g, b};
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
        if (!(rr_abs(v[ch]) <= RR_FLT_MAX)) f |= (v[ch] != v[ch] ? RR_NF_NAN : (v[ch] > 0.0f ? RR_NF_PINF : RR_NF_NINF)) << ch;
    return f;
}
// Adds of one wave to the accumulators, merged: neighbouring lanes that add to the same pixel (the samples of one
// pixel sit in neighbouring lanes, primary_ray; compaction keeps lane order) are summed first with a segmented scan over
// the 16-lane DPP rows, and only the last lane of each run issues the atomics.  Integer adds commute, so the frame
// is the same whatever is merged.  Must be called by all 64 lanes of the wave; lanes with nothing to add pass
// pix = 0xffffffff.  (64 lanes adding to one address, or 16, serialise in the L2 atomic units: +10 ms on sponza_syn.)
#define RR_DPP_SHR(x, n) __builtin_amdgcn_update_dpp(0, (int)(x), 0x110 + (n), 0xf, 0xf, true)
#define RR_DPP_SHL(x, n) __builtin_amdgcn_update_dpp(0, (int)(x), 0x100 + (n), 0xf, 0xf, true)
#define RR_DPP_SHR64(x, n) (((unsigned long long)(uint32_t)RR_DPP_SHR((uint32_t)((x) >> 32), n) << 32) | (uint32_t)RR_DPP_SHR((uint32_t)(x), n))
#define RR_SEG_STEP(n)                                                                                         \
    {                                                                                                          \
        const int pf = RR_DPP_SHR(f, n);                                                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < N; k_++) { const unsigned long long p_ = RR_DPP_SHR64(v[k_], n); if (!f) v[k_] += p_; } \
        if (!f) f = pf;                                                                                        \
    }
// Segmented sum over runs of equal `pix` inside the 16-lane DPP rows: afterwards the LAST lane of every run (return
// value true) holds the run's sums in v[0..N).  Must be called by all 64 lanes; lanes with nothing to add pass 0xffffffff.
template <int N> RR_DEV bool wave_merge_runs(uint32_t pix, unsigned long long (&v)[N]) {
    const uint32_t lane16 = threadIdx.x & 15u;
    const uint32_t prev = (uint32_t)RR_DPP_SHR(pix, 1);
    const int head = (lane16 == 0u || prev != pix) ? 1 : 0; // first lane of a run of equal pixels inside its row
    int f = head;
    RR_SEG_STEP(1) RR_SEG_STEP(2) RR_SEG_STEP(4) RR_SEG_STEP(8)
    const int next_head = RR_DPP_SHL(head, 1);
    return (lane16 == 15u || next_head != 0) && pix != 0xffffffffu;
}
// The same over 32-bit values: when every value of the wave is below 2^25 in magnitude (a colour term up to 2, any
// normal) a run of at most 16 of them sums to less than 2^29, and a step is one DPP add per value instead of a
// 64-bit add with its carry and two moves.
#define RR_SEG_STEP32(n)                                                                                       \
    {                                                                                                          \
        const int pf = RR_DPP_SHR(f, n);                                                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < N; k_++) { const int p_ = RR_DPP_SHR(v[k_], n); if (!f) v[k_] += p_; } \
        if (!f) f = pf;                                                                                        \
    }
template <int N> RR_DEV bool wave_merge_runs32(uint32_t pix, int (&v)[N]) {
    const uint32_t lane16 = threadIdx.x & 15u;
    const uint32_t prev = (uint32_t)RR_DPP_SHR(pix, 1);
    const int head = (lane16 == 0u || prev != pix) ? 1 : 0;
    int f = head;
    RR_SEG_STEP32(1) RR_SEG_STEP32(2) RR_SEG_STEP32(4) RR_SEG_STEP32(8)
    const int next_head = RR_DPP_SHL(head, 1);
    return (lane16 == 15u || next_head != 0) && pix != 0xffffffffu;
}
RR_DEV bool fits25i(int a, int b, int c) { // |a|, |b|, |c| < 2^25
    return (((uint32_t)(a + (1 << 25)) | (uint32_t)(b + (1 << 25)) | (uint32_t)(c + (1 << 25))) >> 26) == 0u;
}
RR_DEV void accum_merged(const DAccum& acc, uint32_t pix, int r, int g, int b) {
    unsigned long long* p = (unsigned long long*)acc.rgb + pix; // one plane per channel
    if (__ballot(!fits25i(r, g, b)) == 0ull) {
        int w[3] = {r, g, b};
        if (wave_merge_runs32<3>(pix, w)) {
            if (w[0]) atomicAdd(p, (unsigned long long)(long long)w[0]);
            if (w[1]) atomicAdd(p + acc.n, (unsigned long long)(long long)w[1]);
            if (w[2]) atomicAdd(p + 2ull * acc.n, (unsigned long long)(long long)w[2]);
        }
        return;
    }
    unsigned long long v[3] = {(unsigned long long)(long long)r, (unsigned long long)(long long)g, (unsigned long long)(long long)b};
    if (wave_merge_runs<3>(pix, v)) {
        if (v[0]) atomicAdd(p, v[0]);
        if (v[1]) atomicAdd(p + acc.n, v[1]);
        if (v[2]) atomicAdd(p + 2ull * acc.n, v[2]);
    }
}
// The aux outputs of the root hits (normal and depth sums, reference src/raytracing.rs:400-402), merged the same way:
// the samples of a pixel sit in neighbouring lanes, and 64 lanes adding to one address serialise in the L2 atomic
// units (measured: k_shade 6.5 -> 42.8 ms on sponza_syn when the four aux adds of every primary hit went out unmerged).
RR_DEV void accum_aux_merged(const DAccum& acc, uint32_t pix, int nx, int ny, int nz, int depth) {
    if (__ballot(!fits25i(nx, ny, nz) || !fits25i(depth, 0, 0)) == 0ull) {
        int w[4] = {nx, ny, nz, depth};
        if (wave_merge_runs32<4>(pix, w)) {
            if (acc.normal) {
                unsigned long long* np = (unsigned long long*)acc.normal + pix;
                if (w[0]) atomicAdd(np, (unsigned long long)(long long)w[0]);
                if (w[1]) atomicAdd(np + acc.n, (unsigned long long)(long long)w[1]);
                if (w[2]) atomicAdd(np + 2ull * acc.n, (unsigned long long)(long long)w[2]);
            }
            if (acc.depth && w[3]) atomicAdd((unsigned long long*)acc.depth + pix, (unsigned long long)(long long)w[3]);
        }
        return;
    }
    unsigned long long v[4] = {(unsigned long long)(long long)nx, (unsigned long long)(long long)ny, (unsigned long long)(long long)nz, (unsigned long long)(long long)depth};
    if (wave_merge_runs<4>(pix, v)) {
        if (acc.normal) {
            unsigned long long* np = (unsigned long long*)acc.normal + pix;
            if (v[0]) atomicAdd(np, v[0]);
            if (v[1]) atomicAdd(np + acc.n, v[1]);
            if (v[2]) atomicAdd(np + 2ull * acc.n, v[2]);
        }
        if (acc.depth && v[3]) atomicAdd((unsigned long long*)acc.depth + pix, v[3]);
    }
}

// Depth terms beyond the 32-bit lane sums (|depth * 2^16| >= 2^25: a root hit farther than 512 units -- a scene modelled in centimetres,
// a far background), merged like everything else: round 3 sent each of them to its accumulator word as an atomic of its own, 64 lanes
// of a level-1 packet to ONE word (the cliff accum_aux_merged's comment describes, for depth alone).  All 64 lanes.
RR_DEV void accum_depth_wide_merged(const DAccum& acc, uint32_t pix, long long depth) {
    unsigned long long v[1] = {(unsigned long long)depth};
    if (wave_merge_runs<1>(pix, v) && acc.depth && v[0]) atomicAdd((unsigned long long*)acc.depth + pix, v[0]);
}

// Share of a launch's packets that is dealt to the waves round-robin, without an atomic (see k_trace_closest).
#ifndef RR_STATIC_NUM
#define RR_STATIC_NUM 1
#define RR_STATIC_DEN 2
#endif

// ---------------------------------------------------------------------------
// primary rays (reference src/raytracing.rs:319-396)
// ---------------------------------------------------------------------------
RR_DEV float4 mat4_mul(const float* m, float x, float y, float z, float w) {
    return make_float4(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w,
                       ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
                       ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w,
                       ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w);
}

// Primary rays are a pure function of (pixel slot, sample): the level-1 trace and shade kernels both DERIVE them
// instead of reading ray records that a ray-generation kernel wrote (round 1: k_raygen wrote 40 B per primary ray,
// 4.7 GB per sponza_syn frame, that the trace kernel read straight back and the shade kernel read a third time).
// Both kernels run the same instruction sequence on the same inputs, so they see the same ray, bit for bit.
struct DPrimary {
    const uint16_t* sample_xy;   // the sub-sample table, samples x (x_i, y_i)
    unsigned long long first;    // first primary index of the batch (sample-major over the region: sample = index / n_pix)
    uint32_t n;                  // primary rays in the batch
    uint32_t group;              // samples of one pixel per 64-ray packet (1 = one sample of 64 pixels)
};
RR_DEV void primary_ray(const DFrame& fr, const uint32_t* __restrict__ slot_xy, const DPrimary& pr, uint32_t i,
                        f3* origin_out, f3* dir_out, uint32_t* pix_out, uint32_t* sample_out) {
    const uint16_t* __restrict__ sample_xy = pr.sample_xy;
    const unsigned long long first = pr.first;
    const uint32_t group = pr.group;
    uint32_t pix, s; // accumulator slot (slots enumerate 8x8 blocks of the region's tiles) and sample
    if (group <= 1u) {
        const unsigned long long gi = first + i; // sample-major index over the region: sample = gi / n_pix
        pix = (uint32_t)(gi % fr.n_region_pixels);
        s = (uint32_t)(gi / fr.n_region_pixels);
    } else {
        // a 64-ray packet = 64/group neighbouring pixels x `group` samples of each (the host guarantees whole groups)
        const uint32_t ppp = RR_WAVE / group;                       // pixels per packet
        const uint32_t packets_per_group = fr.n_region_pixels / ppp;
        const uint32_t pkt = i / RR_WAVE, lane = i % RR_WAVE;
        // the samples of one pixel sit in neighbouring lanes, so that their accumulator adds can be merged (accum_merged)
        pix = (pkt % packets_per_group) * ppp + lane / group;
        s = (uint32_t)(first / fr.n_region_pixels) + (pkt / packets_per_group) * group + lane % group;
    }
    uint32_t xy = slot_xy[pix];
    float x_f = (float)(xy & 0xffffu), y_f = (float)(xy >> 16);
    float w = (float)fr.width, h = (float)fr.height;
    float x_step = 2.0f / w, y_step = 2.0f / h;
    float x_i = (float)sample_xy[2u * s], y_i = (float)sample_xy[2u * s + 1u];
    float inv_cell = 1.0f / (float)fr.cell_size;
    float x_trans = x_step * x_i * inv_cell;
    float y_trans = y_step * y_i * inv_cell;
    if (fr.dof && fr.samples > 1u) { x_trans -= x_step / 2.0f; y_trans -= y_step / 2.0f; }
    f3 origin, dir;
    if (fr.dof) {
        float aperture_scale = (float)fr.width / 800.0f;
        x_trans *= fr.aperture_size * aperture_scale;
        y_trans *= fr.aperture_size * aperture_scale;
        float cx = ((x_f + 0.5f) / w) * 2.0f - 1.0f;
        float cy = 1.0f - ((y_f + 0.5f) / h) * 2.0f;
        float4 cpp = mat4_mul(fr.proj_inv, cx, cy, -1.0f, 1.0f);
        f3 rd = mk3(cpp.x - 0.0f, cpp.y - 0.0f, cpp.z - 0.0f);
        float4 eye = mat4_mul(fr.view_inv, 0.0f, 0.0f, 0.0f, 1.0f);
        float4 dv = mat4_mul(fr.view_inv, rd.x, rd.y, rd.z, 0.0f);
        float dn = sqrtf((dv.x * dv.x + dv.z * dv.z) + (dv.y * dv.y + dv.w * dv.w)); // nalgebra 4-lane dot order
        float4 dvn = make_float4(dv.x / dn, dv.y / dn, dv.z / dn, dv.w / dn);
        float dist = norm3(rd);
        float f = 1.0f / (dist / (dist + fr.focal_length));
        f3 p = mk3(eye.x + f * dvn.x, eye.y + f * dvn.y, eye.z + f * dvn.z);
        float sx = (((x_f + 0.5f) / w) * 2.0f - 1.0f) + x_trans;
        float sy = (1.0f - ((y_f + 0.5f) / h) * 2.0f) + y_trans;
        float4 pp = mat4_mul(fr.proj_inv, sx, sy, -1.0f, 1.0f);
        float4 ro = mat4_mul(fr.view_inv, pp.x, pp.y, pp.z, 1.0f);
        origin = mk3(ro.x, ro.y, ro.z);
        dir = mk3(p.x - ro.x, p.y - ro.y, p.z - ro.z);
    } else {
        float sx = (((x_f + 0.5f) / w) * 2.0f - 1.0f) + x_trans;
        float sy = (1.0f - ((y_f + 0.5f) / h) * 2.0f) + y_trans;
        float4 pp = mat4_mul(fr.proj_inv, sx, sy, -1.0f, 1.0f);
        float4 o = mat4_mul(fr.view_inv, pp.x, pp.y, pp.z, 1.0f);
        float4 d = mat4_mul(fr.view_inv, pp.x - 0.0f, pp.y - 0.0f, pp.z - 0.0f, 0.0f);
        origin = mk3(o.x, o.y, o.z);
        dir = mk3(d.x, d.y, d.z);
    }
    dir = normalize3(dir); // get_color_depth_normal_id normalises on entry (:723)
    *origin_out = origin; *dir_out = dir; *pix_out = pix; *sample_out = s;
}

// ---------------------------------------------------------------------------
// kernel 2: closest hit for a queue of rays
// ---------------------------------------------------------------------------
// Scene view + frame constants of k_shade as one device record.  As by-value kernel arguments all 90 dwords stayed live
// in SGPRs across the kernel's loop and 117 of them were spilled into VGPR lanes; read through a pointer, only what is
// in use is kept (k_shade -4 %).  The trace kernels keep their by-value arguments: their walks use the same few fields
// in every step, and re-reading those costs more than the spills did (closest-hit +4 %, shadow +10 %, measured).
struct DShadeConst { DSceneView sc; DFrame fr; };

// PRIMARY: depth level 1.  The rays are derived from their index (primary_ray), only the 16-B hit record is written;
// block 0 publishes the level's size for the shade kernel and counts the rays.
template <bool PRIMARY>
__global__ __launch_bounds__(RR_BLOCK, RR_CLOSEST_WAVES) void k_trace_closest(DSceneView sc, DRayQueue q, uint32_t* __restrict__ q_count,
                                                            uint32_t* head, const DShadeConst* __restrict__ kc, const uint32_t* __restrict__ slot_xy, DPrimary pr,
                                                            unsigned long long* counters) {
    __shared__ int s_stack[RR_STACK_DEPTH * RR_BLOCK];
    RR_UTIL_KIND(PRIMARY ? 0u : 1u)
    uint32_t n;
    if (PRIMARY) {
        n = pr.n;
        if (blockIdx.x == 0 && threadIdx.x == 0) { *q_count = n; atomicAdd(&counters[RR_CNT_PRIMARY], (unsigned long long)n); }
    } else n = *q_count;
    const uint32_t lane = threadIdx.x & (RR_WAVE - 1);
    // Work distribution.  Locality decides here: the waves that run side by side must walk neighbouring
    // packets (runs of consecutive packets per wave cost 1.8x, measured), and one head word sustains only ~90
    // fetches per microsecond.  So RR_STATIC_NUM/RR_STATIC_DEN of the packets are dealt round-robin with no atomic
    // at all, and the rest is pulled a few packets at a time from the shared head to absorb expensive packets.
    // Blocks b and b + 8 share an XCD (and its L2): with RR_XCD_SWIZZLE the blocks of one XCD take one
    // contiguous run of packets per round instead of every eighth group.
    const uint32_t n_waves = gridDim.x * (RR_BLOCK / RR_WAVE);
    uint32_t blk = blockIdx.x;
#ifndef RR_NO_XCD_SWIZZLE
    if ((gridDim.x & 7u) == 0u) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    const uint32_t wave_id = blk * (RR_BLOCK / RR_WAVE) + threadIdx.x / RR_WAVE;
    const uint32_t n_packets = (n + RR_WAVE - 1) / RR_WAVE;
    // (a launch with few packets per wave -- one rank's share of a frame tiled over 4 or 8 GPUs -- deals three quarters statically:
    // closest-hit 2.36 -> 2.29 ms on a quarter of the contract frame, 1.36 -> 1.32 ms on an eighth; the whole frame loses 5 % to it)
    const bool small_launch = n_packets < 160u * n_waves;
    const uint32_t rounds = (uint32_t)(((unsigned long long)n_packets * (small_launch ? 3u : (uint32_t)RR_STATIC_NUM) / (small_launch ? 4u : (uint32_t)RR_STATIC_DEN)) / n_waves);
    const uint32_t n_static = rounds * n_waves;
    uint32_t round = 0, dyn_next = 0, dyn_left = 0;
    // packets per fetch of the dynamic part: RR_DYN_FETCH on large launches (more costs locality: +4 % at 8, +10 % at 16), fewer when
    // the launch has only a few packets per wave
    const uint32_t dyn_k = n_packets >= 8u * n_waves ? (uint32_t)RR_DYN_FETCH : (n_packets >= 2u * n_waves ? 2u : 1u);
    for (;;) {
        uint32_t pkt;
        if (round < rounds) { pkt = round * n_waves + wave_id; round++; }
        else {
            if (dyn_left == 0u) { // several packets per atomic: the head word sustains only ~90 fetches per microsecond
                uint32_t f = 0;
                if (lane == 0) f = atomicAdd(head, dyn_k);
                dyn_next = n_static + __shfl(f, 0); dyn_left = dyn_k;
            }
            pkt = dyn_next++; dyn_left--;
        }
        if (pkt >= n_packets) break; // wave-uniform
        const uint32_t i = pkt * RR_WAVE + lane;
        const uint32_t ii = min(i, n - 1u); // the lanes past the end of the last packet repeat its last ray, so that the packet form below runs with all lanes
        f3 ro, rd; uint32_t depth;
        if constexpr (PRIMARY) { // the frame constants are read once per packet (not worth 49 SGPRs across the walks); kc is NULL in the <false> build's launches
            uint32_t pix_, smp_; primary_ray(kc->fr, slot_xy, pr, ii, &ro, &rd, &pix_, &smp_); depth = 1u;
        } else {
            const float4 r0 = q.r0[ii], r1 = q.r1[ii];
            ro = mk3(r0.x, r0.y, r0.z); rd = mk3(r1.x, r1.y, r1.z);
            depth = (q.r2[ii].x >> 16) & 0xffu;
        }
        Closest best;
#ifndef RR_NO_BEAM
        if (!trace_closest_packet(sc, ro, rd, depth, s_stack, &best))
#endif
        {
            if (ray_nonfinite(ro, rd)) trace_closest_nonfinite(sc, ro, rd, depth, &best);
            else trace_closest_ray(sc, ro, rd, depth, s_stack, &best);
        }
        if (best.nan_seen) trace_closest_ordered(sc, ro, rd, depth, s_stack, &best); // (rare: a ball whose arithmetic overflows)
        if (i < n) q.hit[i] = make_uint4(__float_as_uint(best.t), (uint32_t)(best.found ? best.item : -1), best.face, 0u);
    }
}

// ---------------------------------------------------------------------------
// wave-level queue allocation: ballot + prefix, one atomic per wave.  Must be
// reached by every lane that may set `want` in the same control-flow region.
// ---------------------------------------------------------------------------
RR_DEV uint32_t wave_alloc(uint32_t* counter, bool want, uint32_t lane) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0u;
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0u;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// get_item_color, reference src/raytracing.rs:677-712
RR_DEV float4 item_color(const DSceneView& sc, const MatR& m, bool has_uv, f2 uv, f3 rgb, int slot) {
    float4 c = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
    float4 t;
    if (tex_color(sc, m, has_uv, uv, slot, &t)) { c.x *= t.x; c.y *= t.y; c.z *= t.z; c.w *= t.w; }
    return c;
}

// ---------------------------------------------------------------------------
// kernel 3: shade one depth level (reference src/raytracing.rs:734-995)
// ---------------------------------------------------------------------------
#ifndef RR_SHADE_WAVES
#define RR_SHADE_WAVES 4
#endif
template <bool PRIMARY>
__global__ __launch_bounds__(RR_BLOCK, RR_SHADE_WAVES) void k_shade(const DShadeConst* __restrict__ kc, const uint32_t* __restrict__ slot_xy, DPrimary pr,
                                                    DRayQueue qin, const uint32_t* __restrict__ qin_count,
                                                    uint32_t chunk_begin, uint32_t chunk_end,
                                                    DRayQueue qout, uint32_t* qout_count,
                                                    DShadowQueue sq, uint32_t* sq_counts, uint32_t sq_segcap, unsigned long long* __restrict__ sq_valid, uint32_t sq_cap,
                                                    DAccum acc, unsigned long long* counters) {
    // The scene view and the frame constants (90 dwords) are read from a small device record where they are needed
    // instead of arriving as kernel arguments: as arguments they were all live in SGPRs across the whole loop and the
    // kernel spilled 117 of them into VGPR lanes.
    const DSceneView& sc = kc->sc;
    const DFrame& fr = kc->fr;
    const uint32_t n = min(*qin_count, chunk_end);
    const uint32_t lane = threadIdx.x & (RR_WAVE - 1);
    const bool gw = sc.general_w != 0u;
    uint32_t n_shaded = 0, n_shadow = 0, n_secondary = 0;
    // the four waves of a workgroup walk four neighbouring packets per iteration and allocate their children together
    // (one atomic per workgroup iteration: a single append counter sustains ~90 returning atomics per microsecond, which
    // made it THE limiter of this kernel on scenes where most hits spawn children)
    __shared__ uint32_t s_child_n[2][RR_BLOCK / RR_WAVE];
    __shared__ float s_lut[256]; // the u8 -> f32 table next to the lanes: a texel decodes with four LDS reads
    s_lut[threadIdx.x] = c_u8_to_f32[threadIdx.x];
    __syncthreads();
    __shared__ uint32_t s_child_base[2];
    const uint32_t wave_in_block = threadIdx.x / RR_WAVE;
    uint32_t parity = 0;
    for (uint32_t block_base = chunk_begin + blockIdx.x * RR_BLOCK; block_base < n; block_base += gridDim.x * RR_BLOCK, parity ^= 1u) {
        const uint32_t base = block_base + wave_in_block * RR_WAVE;
        const uint32_t i = base + lane;
        uint4 hit = make_uint4(0u, 0xffffffffu, 0u, 0u);
        if (i < n) hit = qin.hit[i];
        const int item_idx = (int)hit.y;
        const bool active = i < n && item_idx >= 0; // miss: colour 0, depth 0, normal 0, id 0 (:728-732); buffers are pre-zeroed
        // LEVEL 1: shadow rays have FIXED slots.  The ray of hit i towards the k-th enabled light sits at k * sq_cap +
        // (i - chunk_begin), and one 64-bit word per (light, packet) says which lanes hold one (sq_valid).  A packet of the
        // shadow kernel is then the shadow rays of one packet of hits -- 64 samples of one pixel towards one light, all or
        // none of them as a rule -- which is what its packet form of the top level needs, and nothing is allocated.
        // DEEPER LEVELS: appended densely (ballot + prefix, one atomic per wave and light) to one of RR_SQ_SHARDS sub-queues
        // chosen by the input packet group (256 consecutive rays; a single append counter is a hot word, ~90 returning
        // atomics per microsecond; shard capacity is static: a shard receives at most its share of the chunk's packets).
        // There only some lanes of a packet spawn a shadow ray, and fixed slots would trace packets a fifth full
        // (monkey: shadow 2.1 -> 3.5 ms); so do the level-1 packets of scenes too small for the packet form (+3 .. 15 %).
        const bool sq_fixed = PRIMARY && sq_cap != 0u; // (the host's choice: level 1 of a scene whose top level has a packet form)
        const uint32_t sq_slot = i - chunk_begin;
        uint32_t sq_wrote = 0u; // bit k: this lane wrote a ray for the k-th enabled light
        const uint32_t shard = ((base - chunk_begin) / RR_BLOCK) % RR_SQ_SHARDS; // the 4 packets of a workgroup iteration stay together
        uint32_t* const sq_count = sq_counts + shard * RR_SQ_STRIDE;
        const uint32_t sq_base = shard * sq_segcap;
        int sum_r = 0, sum_g = 0, sum_b = 0; // this hit's direct adds (32-bit fixed point, fix_add), merged with its neighbours' at the end
        // children of this hit (emitted after the hit is shaded, by all waves of the workgroup together)
        bool spawn_refl = false, spawn_refr = false;
        float4 c1_r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1_r1 = c1_r0, c2_r0 = c1_r0, c2_r1 = c1_r0;
        uint2 c1_r2 = make_uint2(0u, 0u), c2_r2 = c1_r2;
        uint32_t sum_pix = 0xffffffffu, aux_pix = 0xffffffffu, nf = 0u, pix_of_nf = 0u;
        int aux_nx = 0, aux_ny = 0, aux_nz = 0, aux_d = 0;
        if (active) {
        n_shaded++;
        float4 r0, r1; uint2 r2;
        if (PRIMARY) { // the root node of a path: throughput 1, depth 1, carries the object id, path node 1
            f3 po, pd; uint32_t ppix, psmp;
            primary_ray(fr, slot_xy, pr, i, &po, &pd, &ppix, &psmp);
            r0 = make_float4(po.x, po.y, po.z, 1.0f);
            r1 = make_float4(pd.x, pd.y, pd.z, __uint_as_float(ppix));
            r2 = make_uint2(psmp | (1u << 16) | (1u << 24), 1u);
        } else { r0 = qin.r0[i]; r1 = qin.r1[i]; r2 = qin.r2[i]; }
        const uint32_t pix = __float_as_uint(r1.w);
        pix_of_nf = pix;
        const uint32_t meta = r2.x, node = r2.y;
        const uint32_t sample = meta & 0xffffu, depth = (meta >> 16) & 0xffu;
        const bool idc = ((meta >> 24) & 1u) != 0u;
        const float thr = r0.w;
        const DItem& it = rr_global(sc.items)[item_idx];
        const uint32_t it_flags = it.flags, it_tri_base = it.tri_base, it_id = it.id; // register copies (see MatR)
        const MatR m = load_material(&rr_global(sc.materials)[it.material], s_lut);
        const f3 ro = mk3(r0.x, r0.y, r0.z), rd = mk3(r1.x, r1.y, r1.z);
        const float hit_dist = __uint_as_float(hit.x);
        const f3 hit_point = ro + (rd * hit_dist);
        const uint32_t slot = hit.z & 0x3fffffffu;
        const bool back = (hit.z >> 31) != 0u, neg = ((hit.z >> 30) & 1u) != 0u;

        // ---- world normal: Shape::intersect (mesh.rs:76-98, sphere.rs:61-65)
        f3 normal;
        DTriAttr at; float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f; bool have_weights = false;
        at.s0 = at.s1 = at.s2 = at.s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (it_flags & RR_IF_SPHERE) {
            LRay lr = inverse_ray(it, ro, rd, gw || ray_nonfinite(ro, rd)); // (see to_local_point)
            float t2 = 0.0f; bool inside = false;
            ray_ball(it.radius, lr, (it_flags & RR_IF_SOLID_BASE) != 0u, &t2, &inside);
            f3 nl = normalize3(lr.o + lr.d * t2);
            normal = to_world_normal(it, inside ? -nl : nl);
        } else {
            const DTri* trp = &rr_global(sc.tris)[it_tri_base + slot];
            // the area weights of the hit point serve the interpolated normal AND the uv (Mesh::get_normal and
            // Mesh::get_uv compute the same three numbers from the same inputs, src/shape/mesh.rs:105-161, :204-259)
            if ((it_flags & RR_IF_SMOOTH) || (m.flags & RR_MF_ANY_TEX)) {
                const float4 v0 = trp->v0, v1 = trp->v1, v2 = trp->v2;
                at = rr_global(sc.attrs)[it_tri_base + slot];
                const f3 p = to_local_point(it, hit_point, gw);
                area_weights(mk3(v0.x, v0.y, v0.z), mk3(v1.x, v1.y, v1.z), mk3(v2.x, v2.y, v2.z), p, v1.w, &a1, &a2, &a3); // v1.w: the triangle's area (host)
                have_weights = true;
            }
            if (it_flags & RR_IF_SMOOTH) {
                f3 p1 = mk3(at.s0.x, at.s0.y, at.s0.z) * a1, p2 = mk3(at.s1.x, at.s1.y, at.s1.z) * a2, p3 = mk3(at.s2.x, at.s2.y, at.s2.z) * a3;
                normal = to_world_normal(it, mk3(p1.x + p2.x + p3.x, p1.y + p2.y + p3.y, p1.z + p2.z + p3.z));
                if (back) normal = -normal;
            } else {
                // to_world_normal(it, neg ? -ng : ng) with ng = DTri::v3, evaluated once per instanced triangle by k_world_normals
                const float4 wn = rr_global(sc.flat_normals)[it.wn_base + 2u * slot + (neg ? 1u : 0u)];
                normal = mk3(wn.x, wn.y, wn.z);
            }
            if (it_flags & RR_IF_FLIP_NORMALS) normal = -normal;
        }
        // ---- aux outputs of the root node (:742-744, :400-402): summed per pixel below, with the wave's other root hits
        if (depth == 1u) {
            aux_pix = pix;
            {   // (a term beyond the 32-bit lane sum is marked and taken in 64 bits with the wave's other wide terms, after the light loop)
                const float dx = hit_dist * RR_DEPTH_SCALE;
                aux_d = rr_abs(dx) < 33554432.0f ? __float2int_rn(dx) : RR_DEPTH_WIDE;
            }
            fix_add(aux_nx, normal.x, RR_FIX_SCALE, RR_FIX_CLAMP, acc.normal, pix);
            fix_add(aux_ny, normal.y, RR_FIX_SCALE, RR_FIX_CLAMP, acc.normal ? acc.normal + acc.n : nullptr, pix);
            fix_add(aux_nz, normal.z, RR_FIX_SCALE, RR_FIX_CLAMP, acc.normal ? acc.normal + 2ull * acc.n : nullptr, pix);
            if ((normal.x - normal.x) + (normal.y - normal.y) + (normal.z - normal.z) + (hit_dist - hit_dist) != 0.0f) { // rare: something is not finite
                if (hit_dist != hit_dist) nf |= RR_NF_DEPTH_NAN;
                if (normal.x != normal.x) nf |= RR_NF_NORMAL_NAN;
                if (normal.y != normal.y) nf |= RR_NF_NORMAL_NAN << 1;
                if (normal.z != normal.z) nf |= RR_NF_NORMAL_NAN << 2;
            }
        }
        // ---- uv (:749-754)
        bool has_uv = false; f2 uv; uv.x = 0.0f; uv.y = 0.0f;
        if (m.flags & RR_MF_ANY_TEX) {
            if (it_flags & RR_IF_SPHERE) uv = sphere_uv(it, hit_point, gw);
            else if (have_weights && (__float_as_uint(at.s3.w) & 1u)) { // Mesh::get_uv with the weights from above
                uv.x = (at.s0.w * a1 + at.s2.w * a2) + at.s3.y * a3;
                uv.y = -((at.s1.w * a1 + at.s3.x * a2) + at.s3.z * a3);
            }
            has_uv = true;
        }
        f3 surface_normal = normal;
        float4 tc;
        // ---- normal mapping (:757-784)
        if (tex_color(sc, m, has_uv, uv, 3, &tc)) {
            f3 tangent = cross3(normal, mk3(0.0f, 1.0f, 0.0f));
            if (norm3(tangent) <= 0.0001f) tangent = cross3(normal, mk3(0.0f, 0.0f, 1.0f));
            tangent = normalize3(tangent);
            f3 bitangent = normalize3(cross3(normal, tangent));
            f3 nm = mk3((tc.x * 2.0f) - 1.0f, (tc.y * 2.0f) - 1.0f, (tc.z * 2.0f) - 1.0f);
            nm.x *= m.normal_map_strength; nm.y *= m.normal_map_strength;
            nm = normalize3(nm);
            f3 t;
            t.x = (tangent.x * nm.x + bitangent.x * nm.y) + normal.x * nm.z;
            t.y = (tangent.y * nm.x + bitangent.y * nm.y) + normal.y * nm.z;
            t.z = (tangent.z * nm.x + bitangent.z * nm.y) + normal.z * nm.z;
            surface_normal = normalize3(t);
        }
        // the generator is keyed on the FRAME pixel (y * width + x), never on the accumulator slot
        const uint32_t xy = slot_xy[pix];
        RngKey rk; rk.seed_lo = fr.seed_lo; rk.seed_hi = fr.seed_hi;
        rk.pixel = (xy >> 16) * fr.width + (xy & 0xffffu); rk.sample = sample; rk.node = node;
        const bool mc = fr.monte_carlo != 0u && (m.flags & RR_MF_MONTE_CARLO) != 0u;
        // ---- roughness (:787-798)
        {
            bool has_rtc = tex_color(sc, m, has_uv, uv, 5, &tc);
            if (mc && (m.roughness > 0.0f || has_rtc)) {
                float roughness = m.roughness, z_lo = m.cos_roughness;
                if (has_rtc) { roughness = (1.0f / RR_PI_F / 2.0f) * tc.x; z_lo = rr_cos(roughness * RR_PI_F); }
                surface_normal = jitter(surface_normal, roughness, z_lo, rk, 0u);
            }
        }
        // ---- colours and alpha (:801-811)
        const float4 ambient_color = item_color(sc, m, has_uv, uv, m.ambient, 1);
        const float4 base_color = item_color(sc, m, has_uv, uv, m.base, 0);
        const float4 specular_color = item_color(sc, m, has_uv, uv, m.specular, 2);
        float alpha = m.alpha * base_color.w;
        if (tex_color(sc, m, has_uv, uv, 4, &tc)) alpha *= tc.x;

        // ---- everything after the light loop that does not depend on it (:922-991)
        float reflectivity = m.reflectivity;
        if (tex_color(sc, m, has_uv, uv, 7, &tc)) reflectivity = tc.x;
        const bool may_recurse = depth <= fr.max_recursion;
        spawn_refl = reflectivity > 0.0f && may_recurse;
        f3 refr_o = mk3(0.0f, 0.0f, 0.0f), refr_d = mk3(0.0f, 0.0f, 0.0f);
        float a_mul = 1.0f; // the factor `color * alpha` applies to what is already in `color`
        if (alpha < 1.0f && may_recurse) {
            // create_transmission (:500-533)
            f3 ref_n = surface_normal;
            float eta_t = m.refraction_index, eta_i = 1.0f;
            float i_dot_n = dot3(rd, surface_normal);
            if (i_dot_n < 0.0f) { i_dot_n = -i_dot_n; }
            else { ref_n = -surface_normal; eta_t = 1.0f; eta_i = m.refraction_index; }
            float eta = eta_i / eta_t;
            float k = 1.0f - (eta * eta) * (1.0f - i_dot_n * i_dot_n);
            if (!(k < 0.0f)) {
                spawn_refr = true;
                refr_o = hit_point + (ref_n * -0.001f);
                refr_d = ((rd + i_dot_n * ref_n) * eta) - (ref_n * sqrtf(k));
                a_mul = alpha;
            }
        } else if (alpha < 1.0f) {
            a_mul = alpha;
        }
        const float fog_amount = rs_min(fr.fog_density * hit_dist, 1.0f);
        float ao = 1.0f;
        if (tex_color(sc, m, has_uv, uv, 6, &tc)) ao = tc.x;
        const float g = (1.0f - fog_amount) * ao;
        const float w_light = thr * ((1.0f - reflectivity) * a_mul * g);
        const float w_refl = thr * (reflectivity * a_mul * g);
        float w_refr = 0.0f; // only a refraction child carries it: fresnel() is evaluated for the hits that spawn one
        if (spawn_refr) {
            const float kr = fresnel(rd, surface_normal, m.refraction_index);
            w_refr = thr * (((kr < 1.0f) ? ((1.0f - kr) * (1.0f - alpha)) : (1.0f - alpha)) * g);
        }

        // The reference multiplies the light sum by (1 - reflectivity), alpha, (1 - fog) and the AO texel AFTER the light
        // loop (:922-991): a NaN in any of them (a map sampled at a NaN uv: acos beyond 1 at a sphere's pole) poisons the
        // pixel even when the light sum is exactly zero or there is no light at all, where no contribution carries it here.
        if (w_light != w_light) nf |= RR_NF_NAN * 7u;
        // ---- constant part: fog colour and ambient / emissive (:977-994)
        {
            float fa = fog_amount * ao;
            sum_pix = pix;
            const float kr_ = thr * (fr.fog_color[0] * fa + ambient_color.x), kg_ = thr * (fr.fog_color[1] * fa + ambient_color.y), kb_ = thr * (fr.fog_color[2] * fa + ambient_color.z);
            nf |= nonfinite_flags(kr_, kg_, kb_);
            fix_add(sum_r, kr_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb, pix); fix_add(sum_g, kg_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + acc.n, pix);
            fix_add(sum_b, kb_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + 2ull * acc.n, pix);
        }
        // ---- object id (:744, :966-969): the last sample's id, passed through fully transparent hits
        const bool child_idc = idc && spawn_refr && approx_equal(alpha, 0.0f);
        if (acc.object_id && idc && !child_idc && sample + 1u == fr.samples) acc.object_id[pix] = it_id;

        // ---- lights (:814-920)
        const f3 view_dir = normalize3(-rd);
        uint32_t lk = 0xffffffffu; // ordinal among the enabled lights
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            const DLight& L = rr_global(sc.lights)[li];
            if (L.type & 0x80u) continue; // disabled
            lk++;
            const f3 lpos = mk3(L.pos[0], L.pos[1], L.pos[2]), ldir = mk3(L.dir[0], L.dir[1], L.dir[2]);
            f3 to_light;
            if (L.type == 0u) to_light = normalize3(-ldir);
            else to_light = normalize3(lpos - hit_point);
            const float dot_light = rs_max(dot3(surface_normal, to_light), 0.0f);
            const f3 ml = -to_light;
            const f3 reflect_dir = ml - ((2.0f * dot3(surface_normal, ml)) * surface_normal);
            const float spec_dot = rs_max(dot3(reflect_dir, view_dir), 0.0f);
            const float light_power = powf(spec_dot, m.shininess);
            float intensity;
            float limit = RR_FLT_MAX;
            if (L.type == 0u) {
                intensity = L.intensity;
            } else {
                const float r2 = norm3(lpos - hit_point);
                limit = r2;
                intensity = L.intensity / (4.0f * RR_PI_F * r2);
                if (L.type == 2u) {
                    const float dd = dot3(-to_light, normalize3(ldir));
                    if (rr_acos(dd) > L.max_angle) intensity = 0.0f;
                }
            }
            const float cr = ((L.color[0] * (specular_color.x * light_power + base_color.x * dot_light)) * intensity) * w_light;
            const float cg = ((L.color[1] * (specular_color.y * light_power + base_color.y * dot_light)) * intensity) * w_light;
            const float cb = ((L.color[2] * (specular_color.z * light_power + base_color.z * dot_light)) * intensity) * w_light;
            const bool nonzero = (cr != 0.0f) || (cg != 0.0f) || (cb != 0.0f);
            // A light term that is exactly zero stays zero whatever the shadow query says, so its ray is not traced -- unless
            // an occluder with an alpha map exists AND this receiver's uv can be non-finite: the reference samples that map at the
            // RECEIVER's uv of the OCCLUDER's hit point (:905), which for a sphere receiver is acos of a value beyond 1 = NaN (for
            // a mesh receiver: the 0 / 0 area weights of a zero-area face), and 0 * NaN = NaN reaches the pixel (found by
            // tools/fuzz_parity.py: a white pixel in the reference, a dark one here).  Every other receiver's uv is finite, the
            // attenuation is finite, and zero stays zero: a scene with alpha-mapped foliage does not pay for the corner case.
            const bool want_shadow = (m.flags & RR_MF_RECEIVE_SHADOW) != 0u && (nonzero || (sc.any_alpha_occluder != 0u && (it_flags & RR_IF_UV_MAY_BE_NAN) != 0u));
            if (!(m.flags & RR_MF_RECEIVE_SHADOW) && nonzero) {
                nf |= nonfinite_flags(cr, cg, cb);
                fix_add(sum_r, cr, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb, pix); fix_add(sum_g, cg, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + acc.n, pix);
                fix_add(sum_b, cb, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + 2ull * acc.n, pix);
            }
            // Hundreds of unshadowed lights: the 32-bit sums are flushed every 32 enabled lights, whether or not THIS light added a term
            // (inside the branch above a light 31 or 63 with a zero term skipped its flush, and 96 terms can pass 2^31).
            if (!(m.flags & RR_MF_RECEIVE_SHADOW) && (lk & 31u) == 31u) {
                if (sum_r) atomicAdd((unsigned long long*)acc.rgb + pix, (unsigned long long)(long long)sum_r);
                if (sum_g) atomicAdd((unsigned long long*)acc.rgb + acc.n + pix, (unsigned long long)(long long)sum_g);
                if (sum_b) atomicAdd((unsigned long long*)acc.rgb + 2ull * acc.n + pix, (unsigned long long)(long long)sum_b);
                sum_r = sum_g = sum_b = 0;
            }
            const uint32_t si = sq_fixed ? lk * sq_cap + sq_slot : sq_base + wave_alloc(sq_count, want_shadow, lane);
            if (want_shadow) {
                if (sq_fixed) sq_wrote |= 1u << lk; // (fixed slots are only chosen for <= RR_FIXED_SLOT_LIGHTS enabled lights)
                f3 so = hit_point + (surface_normal * 0.001f);
                f3 sd = to_light;
                if (mc) sd = jitter(sd, m.shadow_softness, m.cos_shadow_softness, rk, 1u + li);
                sq.s0[si] = make_float4(so.x, so.y, so.z, limit);
                sq.s1[si] = make_float4(sd.x, sd.y, sd.z, __uint_as_float((uint32_t)item_idx | (depth << 27)));
                sq.s2[si] = make_float4(cr, cg, cb, __uint_as_float(pix));
                n_shadow++;
            }
        }

        // ---- children (:938-971): prepared here, compacted into the next level's queue below
        const uint32_t child_meta = sample | ((depth + 1u) << 16);
        if (spawn_refl) {
            // create_reflection (:492-498)
            f3 o2 = hit_point + (surface_normal * 0.001f);
            f3 d2 = normalize3(rd - ((2.0f * dot3(rd, surface_normal)) * surface_normal));
            c1_r0 = make_float4(o2.x, o2.y, o2.z, w_refl);
            c1_r1 = make_float4(d2.x, d2.y, d2.z, __uint_as_float(pix));
            c1_r2 = make_uint2(child_meta, node * 2u);
            n_secondary++;
        }
        if (spawn_refr) {
            f3 d2 = normalize3(refr_d);
            c2_r0 = make_float4(refr_o.x, refr_o.y, refr_o.z, w_refr);
            c2_r1 = make_float4(d2.x, d2.y, d2.z, __uint_as_float(pix));
            c2_r2 = make_uint2(child_meta | (child_idc ? (1u << 24) : 0u), node * 2u + 1u);
            n_secondary++;
        }
        } // active
        if (sq_fixed) { // which lanes of this packet hold a shadow ray, per enabled light (zero words too: the shadow kernel reads them all)
            const uint32_t pk = (base - chunk_begin) / RR_WAVE, n_pk = sq_cap / RR_WAVE;
            for (uint32_t k = 0; k < sc.n_enabled_lights; k++) {
                const unsigned long long mk = __ballot((sq_wrote >> k) & 1u);
                if (lane == 0) sq_valid[k * n_pk + pk] = mk;
            }
        }
        if (nf) atomicOr(&acc.flags[pix_of_nf], nf); // rare: a non-finite term
        accum_merged(acc, sum_pix, sum_r, sum_g, sum_b);
        if ((acc.normal || acc.depth) && __ballot(aux_pix != 0xffffffffu) != 0ull) {
            if (__ballot(aux_d == RR_DEPTH_WIDE) != 0ull) { // hits beyond 512 units: their depth terms in 64 bits, merged per pixel
                long long wide = 0ll;
                if (aux_d == RR_DEPTH_WIDE) { wide = to_fix(__uint_as_float(qin.hit[i].x), RR_DEPTH_SCALE, 1.0e9f); aux_d = 0; }
                accum_depth_wide_merged(acc, wide != 0ll ? aux_pix : 0xffffffffu, wide);
            }
            accum_aux_merged(acc, aux_pix, aux_nx, aux_ny, aux_nz, aux_d);
        }
        {
            // one allocation for all children of the workgroup iteration: per wave the reflection rays first, then the refraction rays
            const unsigned long long m_refl = __ballot(spawn_refl), m_refr = __ballot(spawn_refr);
            const uint32_t n_refl = (uint32_t)__popcll(m_refl), n_both = n_refl + (uint32_t)__popcll(m_refr);
            if (lane == 0) s_child_n[parity][wave_in_block] = n_both;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
                for (uint32_t w2 = 0; w2 < RR_BLOCK / RR_WAVE; w2++) total += s_child_n[parity][w2];
                s_child_base[parity] = total ? atomicAdd(qout_count, total) : 0u;
            }
            __syncthreads();
            uint32_t obase = s_child_base[parity];
            for (uint32_t w2 = 0; w2 < wave_in_block; w2++) obase += s_child_n[parity][w2];
            const unsigned long long below = (1ull << lane) - 1ull;
            if (spawn_refl) {
                const uint32_t oi = obase + (uint32_t)__popcll(m_refl & below);
                qout.r0[oi] = c1_r0; qout.r1[oi] = c1_r1; qout.r2[oi] = c1_r2;
            }
            if (spawn_refr) {
                const uint32_t oi = obase + n_refl + (uint32_t)__popcll(m_refr & below);
                qout.r0[oi] = c2_r0; qout.r1[oi] = c2_r1; qout.r2[oi] = c2_r2;
            }
        }
    }
    // per-wave reduction of the work counters
    for (int off = 32; off > 0; off >>= 1) {
        n_shaded += __shfl_down(n_shaded, off);
        n_shadow += __shfl_down(n_shadow, off);
        n_secondary += __shfl_down(n_secondary, off);
    }
    if (lane == 0) {
        if (n_shaded) atomicAdd(&counters[RR_CNT_SHADED], (unsigned long long)n_shaded);
        if (n_shadow) atomicAdd(&counters[RR_CNT_SHADOW], (unsigned long long)n_shadow);
        if (n_secondary) atomicAdd(&counters[RR_CNT_SECONDARY], (unsigned long long)n_secondary);
    }
}

// The flat world normals of every mesh item (DSceneView::flat_normals): one workgroup per item, both signs of every triangle's
// normal through the item's transform.  Run at scene creation and after rr_scene_update_transforms.
// One workgroup per CHUNK of an item's triangles (`chunks`: (item, first triangle) per workgroup, RR_ITEM_CHUNK triangles each, laid out by
// rr_scene_create: a mesh of a million triangles is 123 workgroups, not one).
#define RR_ITEM_CHUNK 8192u
__global__ __launch_bounds__(RR_BLOCK) void k_world_normals(const DItem* __restrict__ items, const uint2* __restrict__ chunks, const DTri* __restrict__ tris, float4* __restrict__ out) {
    const uint2 ch = chunks[blockIdx.x];
    const DItem& it = items[ch.x];
    if (it.flags & RR_IF_SPHERE) return;
    const uint32_t end = min(it.n_tris, ch.y + RR_ITEM_CHUNK);
    for (uint32_t slot = ch.y + threadIdx.x; slot < end; slot += blockDim.x) {
        const float4 v3 = tris[it.tri_base + slot].v3;
        const f3 ng = mk3(v3.x, v3.y, v3.z);
        const f3 p = to_world_normal(it, ng), m = to_world_normal(it, -ng);
        out[it.wn_base + 2u * slot] = make_float4(p.x, p.y, p.z, 0.0f);
        out[it.wn_base + 2u * slot + 1u] = make_float4(m.x, m.y, m.z, 0.0f);
    }
}

// The extent of every mesh item's SURFACE along the rows of its transform (rr_api.hip: exact_world_box): one workgroup per item over
// the vertices of the mesh's triangles, which are resident (DTri), in double -- per row r the minimum and maximum of
// tr_r.x * p.x + tr_r.y * p.y + tr_r.z * p.z, and the largest |coordinate| per local axis.  Products and sums are IEEE binary64
// without contraction and minimum / maximum are exact, so the host evaluation this replaces (one thread, every vertex of every
// item, three passes: a transform update of sponza_syn spent its time there) gives the same numbers.  A non-finite vertex makes a
// span non-finite; the host then keeps the box of the local box's corners for that item.
// One workgroup per chunk of an item's triangles, as k_world_normals; the host takes the minimum / maximum over an item's chunks.
// out[9 * chunk + (0..2)] = minima, (3..5) = maxima, (6..8) = largest |coordinate|; spheres and empty meshes: +inf / -inf / 0.
__global__ __launch_bounds__(RR_BLOCK) void k_item_spans(const DItem* __restrict__ items, const uint2* __restrict__ chunks, const DTri* __restrict__ tris, double* __restrict__ out) {
    const uint2 ch = chunks[blockIdx.x];
    const DItem& it = items[ch.x];
    const bool mesh = !(it.flags & RR_IF_SPHERE);
    const double inf = __builtin_inf();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, ext[3] = {0.0, 0.0, 0.0};
    bool bad = false; // a NaN coordinate (minimum / maximum would drop it)
    if (mesh) {
        const double m[3][3] = {{(double)it.tr0.x, (double)it.tr0.y, (double)it.tr0.z}, {(double)it.tr1.x, (double)it.tr1.y, (double)it.tr1.z},
                                {(double)it.tr2.x, (double)it.tr2.y, (double)it.tr2.z}};
        const uint32_t end = min(it.n_tris, ch.y + RR_ITEM_CHUNK);
        for (uint32_t slot = ch.y + threadIdx.x; slot < end; slot += RR_BLOCK) {
            const DTri& t = tris[it.tri_base + slot];
            const float4 vs[3] = {t.v0, t.v1, t.v2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double p[3] = {(double)vs[k].x, (double)vs[k].y, (double)vs[k].z};
                if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2]) bad = true;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double v = m[r][0] * p[0] + m[r][1] * p[1] + m[r][2] * p[2];
                    lo[r] = fmin(lo[r], v); hi[r] = fmax(hi[r], v);
                    ext[r] = fmax(ext[r], fabs(p[r]));
                }
            }
        }
    }
    __shared__ double s_red[RR_BLOCK / RR_WAVE][9];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (bad) s_bad = 1;
#pragma unroll
    for (int r = 0; r < 3; r++)
        for (int off = 32; off > 0; off >>= 1) {
            lo[r] = fmin(lo[r], __shfl_down(lo[r], off)); hi[r] = fmax(hi[r], __shfl_down(hi[r], off)); ext[r] = fmax(ext[r], __shfl_down(ext[r], off));
        }
    if ((threadIdx.x & (RR_WAVE - 1)) == 0)
        for (int r = 0; r < 3; r++) { s_red[threadIdx.x / RR_WAVE][r] = lo[r]; s_red[threadIdx.x / RR_WAVE][3 + r] = hi[r]; s_red[threadIdx.x / RR_WAVE][6 + r] = ext[r]; }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int k = (int)threadIdx.x;
        double v = s_red[0][k];
        for (int w = 1; w < RR_BLOCK / RR_WAVE; w++) v = k < 3 ? fmin(v, s_red[w][k]) : fmax(v, s_red[w][k]);
        if (s_bad) v = __builtin_nan("");
        out[9ull * blockIdx.x + k] = v;
    }
}

// ---------------------------------------------------------------------------
// kernel 4: shadow rays of one shade chunk (reference src/raytracing.rs:872-914)
// ---------------------------------------------------------------------------
// FIXED (level 1): packet p = the shadow rays of hit packet (p % packets per light) towards enabled light (p / packets per
// light), slots p * 64 .. p * 64 + 63, lanes by sq_valid[p]; n_packets given.  Otherwise the dense sharded queue of the
// deeper levels (k_shade), n_packets from the shard counts.
template <bool FIXED>
__global__ __launch_bounds__(RR_BLOCK, RR_SHADOW_WAVES) void k_trace_shadow(DSceneView sc, DShadowQueue sq, const uint32_t* __restrict__ sq_counts, uint32_t sq_segcap,
                                                           const unsigned long long* __restrict__ sq_valid, uint32_t n_fixed_packets,
                                                           uint32_t* head, DAccum acc) {
    __shared__ int s_stack[RR_STACK_DEPTH * RR_BLOCK];
    __shared__ uint32_t s_prefix[RR_SQ_SHARDS + 1];
    RR_UTIL_KIND(2u)
    uint32_t n = 0;
    if (!FIXED) { // dense index space over the shards: prefix sums of their counts
        if (threadIdx.x < RR_WAVE) {
            uint32_t c = threadIdx.x < RR_SQ_SHARDS ? sq_counts[threadIdx.x * RR_SQ_STRIDE] : 0u;
            uint32_t incl = c;
            for (int off = 1; off < RR_SQ_SHARDS; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)threadIdx.x >= off) incl += v; }
            if (threadIdx.x < RR_SQ_SHARDS) s_prefix[threadIdx.x + 1] = incl;
            if (threadIdx.x == 0) s_prefix[0] = 0u;
        }
        __syncthreads();
        n = s_prefix[RR_SQ_SHARDS];
    }
    const uint32_t n_packets = FIXED ? n_fixed_packets : (n + RR_WAVE - 1) / RR_WAVE;
    const uint32_t lane = threadIdx.x & (RR_WAVE - 1);
    const bool gw = sc.general_w != 0u;
    // same packet stream as k_trace_closest: most packets dealt round-robin without an atomic (blocks of one XCD
    // take one contiguous run per round), the tail pulled one packet at a time to absorb the expensive ones
    const uint32_t n_waves = gridDim.x * (RR_BLOCK / RR_WAVE);
    uint32_t blk = blockIdx.x;
#ifndef RR_NO_XCD_SWIZZLE
    if ((gridDim.x & 7u) == 0u) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    const uint32_t wave_id = blk * (RR_BLOCK / RR_WAVE) + threadIdx.x / RR_WAVE;
    // (level 1's aligned packets cost much the same and are mostly dealt statically; the dense queue of the deeper levels keeps half dynamic)
    const uint32_t rounds = (uint32_t)(((unsigned long long)n_packets * (FIXED ? RR_SHADOW_FIXED_STATIC_NUM : RR_SHADOW_STATIC_NUM) / (FIXED ? RR_SHADOW_FIXED_STATIC_DEN : RR_SHADOW_STATIC_DEN)) / n_waves);
    const uint32_t n_static = rounds * n_waves;
    uint32_t round = 0, dyn_next = 0, dyn_left = 0;
    // packets per fetch of the dynamic part: RR_DYN_FETCH on large launches (more costs locality: +4 % at 8, +10 % at 16), fewer when
    // the launch has only a few packets per wave
    const uint32_t dyn_k = n_packets >= 8u * n_waves ? (uint32_t)RR_DYN_FETCH : (n_packets >= 2u * n_waves ? 2u : 1u);
    for (;;) {
        uint32_t p;
        if (round < rounds) { p = round * n_waves + wave_id; round++; }
        else {
            if (dyn_left == 0u) {
                uint32_t f = 0;
                if (lane == 0) f = atomicAdd(head, dyn_k);
                dyn_next = n_static + __shfl(f, 0); dyn_left = dyn_k;
            }
            p = dyn_next++; dyn_left--;
        }
        if (p >= n_packets) break;
        unsigned long long valid;
        if (FIXED) valid = sq_valid[p]; // (the same word in every lane)
        else { const uint32_t left = n - p * RR_WAVE; valid = left >= RR_WAVE ? ~0ull : (1ull << left) - 1ull; }
        if (valid == 0ull) continue;
        {
        int sum_r = 0, sum_g = 0, sum_b = 0;
        uint32_t sum_pix = 0xffffffffu;
        const bool live = ((valid >> lane) & 1ull) != 0ull;
        {
            // lanes without a ray repeat the packet's first one (the packet form runs with all lanes) and add nothing
            const uint32_t j = p * RR_WAVE + (live ? lane : (uint32_t)__ffsll((long long)valid) - 1u);
            uint32_t i = j;
            if (!FIXED) {
                uint32_t lo = 0, hi = RR_SQ_SHARDS; // largest shard with prefix <= j
                while (hi - lo > 1u) { uint32_t mid = (lo + hi) >> 1; if (s_prefix[mid] <= j) lo = mid; else hi = mid; }
                i = lo * sq_segcap + (j - s_prefix[lo]);
            }
            const float4 s0 = sq.s0[i];
            const float4 s1 = sq.s1[i], s2 = sq.s2[i];
            const uint32_t rcv_item = __float_as_uint(s1.w) & 0x07ffffffu, rcv_depth = __float_as_uint(s1.w) >> 27;
            const f3 o = mk3(s0.x, s0.y, s0.z), d = mk3(s1.x, s1.y, s1.z);
            ShadowSel sel;
            if (!FIXED || !trace_shadow_packet(sc, o, d, rcv_depth, s0.w, s_stack, &sel)) { if (live) trace_shadow_ray(sc, o, d, rcv_depth, s0.w, s_stack, &sel); }
            if (live) {
            const bool occluded = sel.found && sel.within;
            float factor = 1.0f;
            if (occluded) {
                const DItem& occ = rr_global(sc.items)[sel.item];
                // the RECEIVER's material.alpha (:898); the 2022-05 binary took the occluder's (RR_COMPAT_OCCLUDER_ALPHA_SHADOWS, wave-uniform)
                float shadow_source_alpha = rr_global(sc.materials)[(sc.compat & 1u) ? occ.material : rr_global(sc.items)[rcv_item].material].alpha;
                if (occ.flags & RR_IF_OCCLUDER_ALPHA_TEX) {
                    // the reference evaluates the RECEIVER's get_uv with the occluder's face id (:905)
                    const DItem& rcv = rr_global(sc.items)[rcv_item];
                    const f3 shp = o + (d * sel.t);
                    f2 uv;
                    if (rcv.flags & RR_IF_SPHERE) uv = sphere_uv(rcv, shp, gw);
                    else if (rcv.n_tris != 0u) uv = mesh_uv(sc, rcv, rr_global(sc.face_slot)[rcv.tri_base + sel.face % rcv.n_tris], shp, gw);
                    else { uv.x = 0.0f; uv.y = 0.0f; }
                    float4 tc;
                    if (tex_color(sc, rr_global(sc.materials)[occ.material], true, uv, 4, &tc)) shadow_source_alpha *= tc.x;
                }
                factor = 1.0f - shadow_source_alpha;
            }
            sum_pix = __float_as_uint(s2.w);
            const float vr_ = s2.x * factor, vg_ = s2.y * factor, vb_ = s2.z * factor;
            const uint32_t nf = nonfinite_flags(vr_, vg_, vb_);
            fix_add(sum_r, vr_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb, sum_pix); fix_add(sum_g, vg_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + acc.n, sum_pix);
            fix_add(sum_b, vb_, RR_FIX_SCALE, RR_FIX_CLAMP, acc.rgb + 2ull * acc.n, sum_pix);
            if (nf) atomicOr(&acc.flags[sum_pix], nf);
            }
        }
        accum_merged(acc, sum_pix, sum_r, sum_g, sum_b);
        }
    }
}

// ---------------------------------------------------------------------------
// ray binning between depth levels (counting sort of a level's rays by origin cell x direction octant)
//
// OFF by default (rr_tuning::bin_min_rays).  A deeper level's rays can be re-ordered before they are traced so that a
// wave holds rays that start in the same cell of an 8 x 8 x 8 grid over the level's origins and point into the same
// octant: bounds -> histogram -> prefix -> scatter, 4096 bins.  The order of rays never changes a bit of the frame
// (fixed-point accumulators), so the sort may be unstable: positions inside a bin come from an atomic cursor.
// Measured: it does not pay on this renderer's rays.  Children are spawned in the order of their parents,
// and a packet of parents is 64 samples of ONE pixel: the children of a wave already share their origin to within a
// pixel footprint and differ only by the roughness jitter of their normals (up to +-28 degrees on helmet_syn), which
// no bin of practical size separates.  helmet_syn deeper levels: node steps 21.5 -> 23.1 active lanes of 64, triangle
// tests 11.8 -> 12.9, closest-hit time -2 %, for +2.4 ms of sorting; lotus_syn (refraction, no jitter) gets slower.
// ---------------------------------------------------------------------------
#define RR_BIN_AXIS 8u
#define RR_BIN_COUNT (RR_BIN_AXIS * RR_BIN_AXIS * RR_BIN_AXIS * 8u)
RR_DEV int float_ordered(float f) { const int i = __float_as_int(f); return i ^ ((i >> 31) & 0x7fffffff); } // monotone map f32 -> i32
RR_DEV float ordered_float(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

// bounds[0..2] = min, bounds[3..5] = max of the finite ray origins, as ordered ints (host presets +max / -max)
__global__ __launch_bounds__(RR_BLOCK) void k_bin_bounds(DRayQueue q, uint32_t n, int* bounds) {
    float lo[3] = {RR_FLT_MAX, RR_FLT_MAX, RR_FLT_MAX}, hi[3] = {-RR_FLT_MAX, -RR_FLT_MAX, -RR_FLT_MAX};
    for (uint32_t i = blockIdx.x * RR_BLOCK + threadIdx.x; i < n; i += gridDim.x * RR_BLOCK) {
        const float4 r0 = q.r0[i];
        const float o[3] = {r0.x, r0.y, r0.z};
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (rr_abs(o[k]) <= RR_FLT_MAX) { lo[k] = fminf(lo[k], o[k]); hi[k] = fmaxf(hi[k], o[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off)); }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&bounds[k], float_ordered(lo[k])); atomicMax(&bounds[3 + k], float_ordered(hi[k])); }
    }
}
RR_DEV uint32_t bin_cell(float o, float lo, float inv_step) { // NaN -> 0
    const float c = fminf(fmaxf((o - lo) * inv_step, 0.0f), (float)(RR_BIN_AXIS - 1u));
    return (uint32_t)c;
}
RR_DEV uint32_t spread3(uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4); } // 3 bits -> every third bit
RR_DEV uint32_t bin_key(float4 r0, float4 r1, const int* __restrict__ bounds) {
    const float lx = ordered_float(bounds[0]), ly = ordered_float(bounds[1]), lz = ordered_float(bounds[2]);
    const float ex = ordered_float(bounds[3]) - lx, ey = ordered_float(bounds[4]) - ly, ez = ordered_float(bounds[5]) - lz;
    const float sx = ex > 0.0f ? (float)RR_BIN_AXIS / ex : 0.0f, sy = ey > 0.0f ? (float)RR_BIN_AXIS / ey : 0.0f, sz = ez > 0.0f ? (float)RR_BIN_AXIS / ez : 0.0f;
    const uint32_t cell = spread3(bin_cell(r0.x, lx, sx)) | (spread3(bin_cell(r0.y, ly, sy)) << 1) | (spread3(bin_cell(r0.z, lz, sz)) << 2); // Morton order
    const uint32_t oct = (__float_as_uint(r1.x) >> 31) | ((__float_as_uint(r1.y) >> 31) << 1) | ((__float_as_uint(r1.z) >> 31) << 2);
    return (cell << 3) | oct;
}
// One add per DISTINCT key of a wave instead of one per lane (neighbouring rays mostly share their bin, and 64 lanes
// adding to one word serialise): the lanes are peeled off key by key; returns this lane's slot when `ret`.
RR_DEV uint32_t wave_add_by_key(uint32_t* table, uint32_t key, bool valid, bool ret) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t slot = 0u;
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, leader);
        const unsigned long long same = __ballot(valid && key == k);
        uint32_t base = 0u;
        if ((int)lane == leader) base = atomicAdd(&table[k], (uint32_t)__popcll(same));
        if (ret) { base = (uint32_t)__shfl((int)base, leader); if (valid && key == k) slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); }
        todo &= ~same;
    }
    return slot;
}
// keys into the (still unused) hit records of the level, histogram through LDS
__global__ __launch_bounds__(RR_BLOCK) void k_bin_count(DRayQueue q, uint32_t n, const int* __restrict__ bounds, uint32_t* hist) {
    __shared__ uint32_t s_hist[RR_BIN_COUNT];
    for (uint32_t b = threadIdx.x; b < RR_BIN_COUNT; b += RR_BLOCK) s_hist[b] = 0u;
    __syncthreads();
    const uint32_t n_up = ((n + RR_BLOCK - 1) / RR_BLOCK) * RR_BLOCK; // whole waves take part in the peeling loop
    for (uint32_t i = blockIdx.x * RR_BLOCK + threadIdx.x; i < n_up; i += gridDim.x * RR_BLOCK) {
        uint32_t key = 0u;
        if (i < n) { key = bin_key(q.r0[i], q.r1[i], bounds); q.hit[i] = make_uint4(key, 0u, 0u, 0u); }
        wave_add_by_key(s_hist, key, i < n, false);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < RR_BIN_COUNT; b += RR_BLOCK) { const uint32_t c = s_hist[b]; if (c) atomicAdd(&hist[b], c); }
}
// exclusive prefix of the histogram, in place (one workgroup of 1024 threads, 4 bins each)
__global__ __launch_bounds__(1024) void k_bin_prefix(uint32_t* hist) {
    __shared__ uint32_t s_part[1024];
    const uint32_t t = threadIdx.x;
    uint32_t v[RR_BIN_COUNT / 1024], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < RR_BIN_COUNT / 1024; k++) { v[k] = hist[t * (RR_BIN_COUNT / 1024) + k]; sum += v[k]; }
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        const uint32_t add = t >= off ? s_part[t - off] : 0u;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;
#pragma unroll
    for (uint32_t k = 0; k < RR_BIN_COUNT / 1024; k++) { hist[t * (RR_BIN_COUNT / 1024) + k] = run; run += v[k]; }
}
// every ray takes the next free slot of its bin (`cursor` = the exclusive prefix, advanced atomically)
__global__ __launch_bounds__(RR_BLOCK) void k_bin_scatter(DRayQueue src, DRayQueue dst, uint32_t n, uint32_t* cursor) {
    const uint32_t n_up = ((n + RR_BLOCK - 1) / RR_BLOCK) * RR_BLOCK;
    for (uint32_t i = blockIdx.x * RR_BLOCK + threadIdx.x; i < n_up; i += gridDim.x * RR_BLOCK) {
        const uint32_t pos = wave_add_by_key(cursor, i < n ? src.hit[i].x : 0u, i < n, true);
        if (i < n) { dst.r0[pos] = src.r0[i]; dst.r1[pos] = src.r1[i]; dst.r2[pos] = src.r2[i]; }
    }
}

// ---------------------------------------------------------------------------
// kernel 5: resolve accumulators into PixelData (reference src/raytracing.rs:406-426)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RR_BLOCK) void k_resolve(DFrame fr, const uint32_t* __restrict__ slot_xy,
                                                      const uint32_t* __restrict__ slot_out, DAccum acc,
                                                      uint8_t* rgba8, float* normal, float* depth, uint32_t* object_id,
                                                      uint32_t frame_layout) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; // accumulator slot
    if (p >= fr.n_region_pixels) return;
    uint32_t o;
    if (frame_layout) { uint32_t xy = slot_xy[p]; o = (xy >> 16) * fr.width + (xy & 0xffffu); }
    else o = slot_out[p]; // position in the region's compact output order
    const double inv_fix = 1.0 / 16777216.0;
    const float n = (float)fr.samples;
    const uint32_t nf = acc.flags[p];
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float sum = (float)((double)acc.rgb[(unsigned long long)k * acc.n + p] * inv_fix);
        // what the reference's f32 sum would hold if a sample was not finite: NaN (also +inf + -inf) or +-inf
        const bool pinf = (nf >> (3 + k)) & 1u, ninf = (nf >> (6 + k)) & 1u;
        if (((nf >> k) & 1u) || (pinf && ninf)) sum = __builtin_nanf("");
        else if (pinf) sum = __builtin_inff();
        else if (ninf) sum = -__builtin_inff();
        float v = sum / n;
        c[k] = rs_min(v, 1.0f); // f32::min: NaN.min(1.0) = 1.0
    }
    uint32_t r, g, b;
    if (fr.gamma) {
        const float ig = 1.0f / 2.2f;
        r = as_u8(powf(c[0], ig) * 255.0f); g = as_u8(powf(c[1], ig) * 255.0f); b = as_u8(powf(c[2], ig) * 255.0f);
    } else {
        r = as_u8(c[0] * 255.0f); g = as_u8(c[1] * 255.0f); b = as_u8(c[2] * 255.0f);
    }
    ((uint32_t*)rgba8)[o] = r | (g << 8) | (b << 16) | (255u << 24);
    if (normal && acc.normal) {
        f3 nn = mk3((float)((double)acc.normal[p] * inv_fix) / n, (float)((double)acc.normal[acc.n + p] * inv_fix) / n,
                    (float)((double)acc.normal[2ull * acc.n + p] * inv_fix) / n);
        if (nf & (RR_NF_NORMAL_NAN * 7u)) { // a NaN sample normal poisons its component, and through the norm all three
            if (nf & RR_NF_NORMAL_NAN) nn.x = __builtin_nanf("");
            if (nf & (RR_NF_NORMAL_NAN << 1)) nn.y = __builtin_nanf("");
            if (nf & (RR_NF_NORMAL_NAN << 2)) nn.z = __builtin_nanf("");
        }
        nn = normalize3(nn); // 0/0 = NaN on all-miss pixels, as in the reference (:426)
        normal[3ull * o] = nn.x; normal[3ull * o + 1] = nn.y; normal[3ull * o + 2] = nn.z;
    }
    if (depth && acc.depth) depth[o] = (nf & RR_NF_DEPTH_NAN) ? __builtin_nanf("") : (float)((double)acc.depth[p] * (1.0 / 65536.0)) / n;
    if (object_id && acc.object_id) object_id[o] = acc.object_id[p];
}

// ---------------------------------------------------------------------------
// kernel 6: gather compact per-rank buffers into frame order (multi-GPU epilogue)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RR_BLOCK) void k_gather_frame(const uint32_t* __restrict__ src_index, uint32_t n_pixels,
                                                           uint32_t elem_words, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels * elem_words) return;
    uint32_t p = i / elem_words, w = i % elem_words;
    dst[i] = src[(uint64_t)src_index[p] * elem_words + w];
}

// The same for the gathered PACKS of a multi-rank frame (TiledFrame: every rank's byte buffer holds one section per output buffer,
// each section sized for the largest rank): all four buffers in ONE launch, straight from the gather target -- no concatenation
// pass, no launch per buffer.  src_rank / src_local: for every frame pixel, the rank that rendered it and its position in that
// rank's compact order.  A thread moves one 4-byte word; the words of a pixel are (rgba, normal.xyz, depth, object id) = 6.
struct DPackedGather {
    const uint32_t* src_rank; const uint32_t* src_local;
    const char* packs; unsigned long long pack_stride;   // rank r's pack starts at packs + r * pack_stride
    unsigned long long section[4];                       // byte offset of each buffer's section inside a pack
    uint32_t words[4];                                   // 4-byte words per pixel of each buffer (0 = absent)
    uint32_t* dst[4];
    uint32_t n_pixels, words_total;
};
__global__ __launch_bounds__(RR_BLOCK) void k_gather_packed(DPackedGather g) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned long long)g.n_pixels * g.words_total) return;
    const uint32_t p = (uint32_t)(i / g.words_total);
    uint32_t w = (uint32_t)(i % g.words_total);
    uint32_t b = 0;
    while (w >= g.words[b]) { w -= g.words[b]; b++; } // (absent buffers have 0 words: skipped)
    const uint32_t* src = (const uint32_t*)(g.packs + (unsigned long long)g.src_rank[p] * g.pack_stride + g.section[b]);
    g.dst[b][(unsigned long long)p * g.words[b] + w] = src[(unsigned long long)g.src_local[p] * g.words[b] + w];
}

// ---------------------------------------------------------------------------
// kernel 7: post-processing (reference src/post_processing.rs:24-181), one thread per pixel.
// HBM-bound: 24 algorithmic bytes per pixel (RGBA in + normal + object id + RGBA out; the 4-neighbour
// re-reads come from L1/L2).  Neighbours are addressed by LINEAR index y * width + x and only the linear index is
// range-checked (:40-45), so x + 1 at the right border reads the first pixel of the next row, as in the reference.
// ---------------------------------------------------------------------------
RR_DEV float curvature_soft_clamp(float curvature, float control) {
    if (curvature < 0.5f / control) return curvature * (1.0f - curvature * control);
    return 0.25f / control;
}
__global__ __launch_bounds__(RR_BLOCK) void k_post_process(uint32_t width, uint32_t height, uint32_t cavity, uint32_t outline,
                                                           const uint32_t* __restrict__ rgba_in, const float* __restrict__ normal,
                                                           const uint32_t* __restrict__ object_id, uint32_t* __restrict__ rgba_out) {
    const long long n = (long long)width * height;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t px = rgba_in[i];
    float r = (float)(px & 255u), g = (float)((px >> 8) & 255u), b = (float)((px >> 16) & 255u);
    const long long w = (long long)width;
    const long long up = i + w, down = i - w, left = i - 1, right = i + 1; // (0,1) (0,-1) (-1,0) (1,0)
    if (outline) {
        const uint32_t c = object_id[i];
        const uint32_t o_up = (up >= 0 && up < n) ? object_id[up] : 0u, o_down = (down >= 0 && down < n) ? object_id[down] : 0u;
        const uint32_t o_l = (left >= 0 && left < n) ? object_id[left] : 0u, o_r = (right >= 0 && right < n) ? object_id[right] : 0u;
        // equal_vec . (0.25, 0.25, 0.25, 0.25) in nalgebra's 4-lane order (x*x' + z*z') + (y*y' + w*w'); lanes = (up, down, -1, +1)
        const float e0 = (o_up == c) ? 1.0f : 0.0f, e1 = (o_down == c) ? 1.0f : 0.0f, e2 = (o_l == c) ? 1.0f : 0.0f, e3 = (o_r == c) ? 1.0f : 0.0f;
        const float opacity = 1.0f - ((e0 * 0.25f + e2 * 0.25f) + (e1 * 0.25f + e3 * 0.25f));
        if (opacity > 0.0f) { r = opacity * 255.0f; g = opacity * 255.0f; b = opacity * 255.0f; }
    }
    if (cavity) {
        // .xz() of the neighbour normals: x -> .x, z -> .y
        const float up_z = (up >= 0 && up < n) ? normal[3 * up + 2] : 0.0f, down_z = (down >= 0 && down < n) ? normal[3 * down + 2] : 0.0f;
        const float left_x = (left >= 0 && left < n) ? normal[3 * left] : 0.0f, right_x = (right >= 0 && right < n) ? normal[3 * right] : 0.0f;
        const float diff = (up_z - down_z) + (right_x - left_x);
        float curvature;
        if (diff < 0.0f) curvature = -2.0f * curvature_soft_clamp(-diff, 1.0f);
        else curvature = 2.0f * curvature_soft_clamp(diff, 1.15f);
        r *= curvature + 1.0f; g *= curvature + 1.0f; b *= curvature + 1.0f;
    }
    // f32::clamp(0, 255) keeps NaN, `as u8` maps NaN to 0
    r = (r < 0.0f) ? 0.0f : ((r > 255.0f) ? 255.0f : r);
    g = (g < 0.0f) ? 0.0f : ((g > 255.0f) ? 255.0f : g);
    b = (b < 0.0f) ? 0.0f : ((b > 255.0f) ? 255.0f : b);
    rgba_out[i] = as_u8(r) | (as_u8(g) << 8) | (as_u8(b) << 16) | (255u << 24);
}

// ---------------------------------------------------------------------------
// kernel 8: device self-test of the arithmetic contract (tests/test_device_math.py)
// op: 0 sincos -> (sin, cos); 1 acos; 2 atan2(a, b); 3 a / b; 4 sqrt(a); 5 jitter(dir = (a, b, c)); 7 cos(a * pi) as jitter() needs it
// (op 6 is the HOST build of the same rr_cos, rr_api.hip)
// ---------------------------------------------------------------------------
__global__ void k_math_probe(int op, const float* a, const float* b, const float* c, int n, float* out0, float* out1, float* out2,
                             uint32_t seed_lo, uint32_t seed_hi) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 0) { float s, co; rr_sincos(a[i], &s, &co); out0[i] = s; out1[i] = co; }
    else if (op == 1) out0[i] = rr_acos(a[i]);
    else if (op == 2) out0[i] = rr_atan2(a[i], b[i]);
    else if (op == 3) out0[i] = a[i] / b[i];
    else if (op == 4) out0[i] = sqrtf(a[i]);
    else if (op == 7) out0[i] = rr_cos(a[i] * RR_PI_F);
    else if (op == 10) { // the per-triangle shading constants as k_shade evaluated them per hit (now DTri::v1.w, v3; op 11 is the host's build)
        const int t = (i / 3) * 3, k = i % 3;
        if (t + 2 < n) {
            const f3 va = mk3(a[t], a[t + 1], a[t + 2]), vb = mk3(b[t], b[t + 1], b[t + 2]), vc = mk3(c[t], c[t + 1], c[t + 2]);
            const f3 ng = normalize3(cross3(vb - va, vc - va));
            out0[i] = k == 0 ? ng.x : (k == 1 ? ng.y : ng.z);
            out1[i] = norm3(cross3(va - vb, va - vc));
        }
    }
    else if (op == 5) {
        RngKey k; k.seed_lo = seed_lo; k.seed_hi = seed_hi; k.pixel = (uint32_t)i; k.sample = (uint32_t)(i & 7); k.node = 1u + (uint32_t)(i % 5);
        f3 r = jitter(mk3(a[i], b[i], c[i]), 0.05f, rr_cos(0.05f * RR_PI_F), k, (uint32_t)(i % 3));
        out0[i] = r.x; out1[i] = r.y; out2[i] = r.z;
    }
}
