// rr_bvh.cpp — host-side BVH2 builder for the device layout of rr_device.h.
//
// Stands in for the two acceleration structures of the reference: parry3d's
// per-TriMesh Qbvh (reference src/shape/mesh.rs:171) and the `bvh` crate's
// scene BVH (reference src/scene.rs:1674-1688).  Only the SET of primitives a
// traversal reaches matters for the result, so the tree shape is free: binned
// SAH (16 bins, 3 axes), leaves of up to RR_MAX_LEAF_TRIS primitives, and a
// hard depth limit (object-median splits once the remaining depth budget is
// just enough) so that the fixed-size LDS traversal stack can never overflow.
#include "rr_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rr {

namespace {

struct Prim { float lo[3], hi[3], c[3]; uint32_t id; };

struct Builder {
    std::vector<Prim> prims;
    std::vector<DNode>* nodes;
    uint32_t max_leaf;
    int max_depth;
    int depth_reached = 0;

    static float half_area(const float* lo, const float* hi) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    void bounds(uint32_t a, uint32_t b, float* lo, float* hi) const {
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (uint32_t i = a; i < b; i++)
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], prims[i].lo[k]); hi[k] = std::max(hi[k], prims[i].hi[k]); }
    }
    // Padding keeps the conservative slab test of the kernels from ever culling a
    // primitive that the exact (parry-style) test would accept.
    static void pad(float* lo, float* hi) {
        for (int k = 0; k < 3; k++) {
            float m = std::max(std::fabs(lo[k]), std::fabs(hi[k]));
            float e = m * 4.0e-6f + (hi[k] - lo[k]) * 4.0e-6f + 1.0e-30f;
            lo[k] -= e; hi[k] += e;
        }
    }
    static int32_t leaf_code(uint32_t first, uint32_t count) { return ~(int32_t)(first | ((count - 1u) << 28)); }
    // levels a perfectly balanced split of n primitives still needs below this node
    int levels_needed(uint32_t n) const {
        int l = 0;
        uint32_t cap = max_leaf;
        while (cap < n) { cap *= 2u; l++; }
        return l;
    }

    int32_t build(uint32_t a, uint32_t b, int depth) {
        uint32_t n = b - a;
        depth_reached = std::max(depth_reached, depth);
        if (n <= 1u) return leaf_code(a, n ? n : 1u);
        int budget = max_depth - depth; // inner levels still allowed below (including this one)
        bool force_median = levels_needed(n) >= budget;
        uint32_t mid = 0;
        bool split_found = false;
        if (!force_median) {
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = a; i < b; i++)
                for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], prims[i].c[k]); chi[k] = std::max(chi[k], prims[i].c[k]); }
            const int NB = 16;
            float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                float ext = chi[ax] - clo[ax];
                if (!(ext > 0.0f)) continue;
                float blo[NB][3], bhi[NB][3]; uint32_t bc[NB];
                for (int i = 0; i < NB; i++) { bc[i] = 0; for (int k = 0; k < 3; k++) { blo[i][k] = INFINITY; bhi[i][k] = -INFINITY; } }
                float scale = (float)NB / ext;
                for (uint32_t i = a; i < b; i++) {
                    int bi = std::min(NB - 1, std::max(0, (int)((prims[i].c[ax] - clo[ax]) * scale)));
                    bc[bi]++;
                    for (int k = 0; k < 3; k++) { blo[bi][k] = std::min(blo[bi][k], prims[i].lo[k]); bhi[bi][k] = std::max(bhi[bi][k], prims[i].hi[k]); }
                }
                float ra[NB]; uint32_t rc[NB];
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}; uint32_t cnt = 0;
                for (int i = NB - 1; i >= 1; i--) {
                    cnt += bc[i];
                    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[i][k]); hi[k] = std::max(hi[k], bhi[i][k]); }
                    ra[i] = cnt ? half_area(lo, hi) : 0.0f; rc[i] = cnt;
                }
                for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
                cnt = 0;
                for (int i = 0; i < NB - 1; i++) {
                    cnt += bc[i];
                    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[i][k]); hi[k] = std::max(hi[k], bhi[i][k]); }
                    if (cnt == 0 || rc[i + 1] == 0) continue;
                    float cost = half_area(lo, hi) * (float)cnt + ra[i + 1] * (float)rc[i + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
                }
            }
            if (best_axis >= 0) {
                if (n <= max_leaf) {
                    float plo[3], phi[3];
                    bounds(a, b, plo, phi);
                    // cost of one more traversal step, in primitive tests (RR_SAH_CT overrides, for experiments)
                    static const float ct = getenv("RR_SAH_CT") ? (float)atof(getenv("RR_SAH_CT")) : 1.5f;
                    if (best_cost + ct * half_area(plo, phi) >= half_area(plo, phi) * (float)n) return leaf_code(a, n);
                }
                float ext = chi[best_axis] - clo[best_axis];
                float scale = 16.0f / ext;
                float c0 = clo[best_axis];
                int ax = best_axis, bb = best_bin;
                auto it = std::partition(prims.begin() + a, prims.begin() + b, [&](const Prim& p) {
                    int bi = std::min(15, std::max(0, (int)((p.c[ax] - c0) * scale)));
                    return bi <= bb;
                });
                mid = (uint32_t)(it - prims.begin());
                split_found = mid > a && mid < b;
            } else if (n <= max_leaf) {
                return leaf_code(a, n);
            }
        }
        if (!split_found) {
            if (n <= max_leaf && !force_median) return leaf_code(a, n);
            if (n <= max_leaf && force_median) return leaf_code(a, n);
            // object median along the longest centroid axis
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = a; i < b; i++)
                for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], prims[i].c[k]); chi[k] = std::max(chi[k], prims[i].c[k]); }
            int ax = 0;
            if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
            if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
            mid = a + n / 2u;
            std::nth_element(prims.begin() + a, prims.begin() + mid, prims.begin() + b,
                             [ax](const Prim& p, const Prim& q) { return p.c[ax] < q.c[ax] || (p.c[ax] == q.c[ax] && p.id < q.id); });
        }
        int32_t idx = (int32_t)nodes->size();
        nodes->emplace_back();
        int32_t l = build(a, mid, depth + 1);
        int32_t r = build(mid, b, depth + 1);
        float l0[3], h0[3], l1[3], h1[3];
        bounds(a, mid, l0, h0); pad(l0, h0);
        bounds(mid, b, l1, h1); pad(l1, h1);
        DNode& nd = (*nodes)[idx];
        nd.n0 = make_float4(l0[0], h0[0], l0[1], h0[1]);
        nd.n1 = make_float4(l1[0], h1[0], l1[1], h1[1]);
        nd.n2 = make_float4(l0[2], h0[2], l1[2], h1[2]);
        float fl, fr;
        std::memcpy(&fl, &l, 4); std::memcpy(&fr, &r, 4);
        nd.n3 = make_float4(fl, fr, 0.0f, 0.0f);
        return idx;
    }
};

} // namespace

bool build_bvh(const float* boxes_lo, const float* boxes_hi, uint32_t n, uint32_t max_leaf, int max_depth, BvhResult* out) {
    Builder b;
    b.prims.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        Prim& p = b.prims[i];
        p.id = i;
        for (int k = 0; k < 3; k++) {
            p.lo[k] = boxes_lo[3 * (size_t)i + k]; p.hi[k] = boxes_hi[3 * (size_t)i + k];
            p.c[k] = 0.5f * (p.lo[k] + p.hi[k]);
        }
    }
    out->nodes.clear();
    out->nodes.reserve(n);
    b.nodes = &out->nodes;
    b.max_leaf = max_leaf;
    b.max_depth = max_depth;
    out->root = n ? b.build(0, n, 0) : 0;
    out->order.resize(n);
    for (uint32_t i = 0; i < n; i++) out->order[i] = b.prims[i].id;
    out->depth = b.depth_reached;
    return b.depth_reached <= max_depth;
}

} // namespace rr
