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
#include <limits>

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
                    const float ct = 1.5f; // cost of one more traversal step, in primitive tests (0.5 - 2 measured: +-1 %)
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

namespace {
struct Box4 { float lo[3], hi[3]; };
inline int32_t child_of(const DNode& n, int k) { int32_t c; std::memcpy(&c, k == 0 ? &n.n3.x : &n.n3.y, 4); return c; }
inline Box4 box_of(const DNode& n, int k) {
    Box4 b;
    if (k == 0) { b.lo[0] = n.n0.x; b.hi[0] = n.n0.y; b.lo[1] = n.n0.z; b.hi[1] = n.n0.w; b.lo[2] = n.n2.x; b.hi[2] = n.n2.y; }
    else { b.lo[0] = n.n1.x; b.hi[0] = n.n1.y; b.lo[1] = n.n1.z; b.hi[1] = n.n1.w; b.lo[2] = n.n2.z; b.hi[2] = n.n2.w; }
    return b;
}
struct Collapse {
    const BvhResult& b2;
    std::vector<int> height; // BVH2 inner levels below (and including) each node
    std::vector<DNode4>* out;
    int limit, max_pending;
    bool greedy;
    int h_of(int32_t code) const { return code < 0 ? 0 : height[code]; }
    int fill_heights(int32_t n) {
        if (n < 0) return 0;
        int h = 1 + std::max(fill_heights(child_of(b2.nodes[n], 0)), fill_heights(child_of(b2.nodes[n], 1)));
        height[n] = h;
        return h;
    }
    // A walk that is `pending` entries deep when it reaches n2 must still fit the stack below it.  A node adopts its
    // grandchildren only when every resulting subtree keeps that promise in the worst case (binary nodes all the way
    // down cost one entry per BVH2 level), so trees up to `limit` BVH2 levels stay traversable with `limit` entries.
    int32_t rec(int32_t n2, int pending) {
        if (n2 < 0) { max_pending = std::max(max_pending, pending); return n2; }
        struct Slot { int32_t code; Box4 box; };
        const int32_t ch[2] = {child_of(b2.nodes[n2], 0), child_of(b2.nodes[n2], 1)};
        Slot slots[4]; int ns = 0;
        if (!greedy) {
            // open both children, one, or none: the first plan whose subtrees all fit the stack budget
            static const int plans[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
            for (int p = 0; p < 4; p++) {
                ns = 0;
                bool ok = true;
                for (int k = 0; k < 2; k++) {
                    if (plans[p][k] && ch[k] >= 0) {
                        for (int g = 0; g < 2; g++) { slots[ns].code = child_of(b2.nodes[ch[k]], g); slots[ns].box = box_of(b2.nodes[ch[k]], g); ns++; }
                    } else { slots[ns].code = ch[k]; slots[ns].box = box_of(b2.nodes[n2], k); ns++; }
                }
                for (int s = 0; s < ns; s++) ok = ok && (pending + (ns - 1) + h_of(slots[s].code) <= limit);
                if (ok) break;
            }
        } else {
            // Greedy by surface area: while a slot is free, open the inner child with the largest box (this may go three
            // BVH2 levels down on one side), as long as every resulting subtree still fits the stack budget.
            for (int k = 0; k < 2; k++) { slots[ns].code = ch[k]; slots[ns].box = box_of(b2.nodes[n2], k); ns++; }
            auto area = [](const Box4& bx) { const float dx = bx.hi[0] - bx.lo[0], dy = bx.hi[1] - bx.lo[1], dz = bx.hi[2] - bx.lo[2]; return dx * dy + dy * dz + dz * dx; };
            auto fits = [&](const Slot* sl, int n) { for (int s = 0; s < n; s++) if (pending + (n - 1) + h_of(sl[s].code) > limit) return false; return true; };
            bool closed[4] = {false, false, false, false}; // a slot that could not be opened stays as it is
            while (ns < 4) {
                int pick = -1; float best = -1.0f;
                for (int s = 0; s < ns; s++)
                    if (slots[s].code >= 0 && !closed[s]) { const float ar = area(slots[s].box); if (ar > best) { best = ar; pick = s; } }
                if (pick < 0) break;
                Slot trial[4]; int nt = 0;
                for (int s = 0; s < ns; s++) {
                    if (s == pick) { for (int g = 0; g < 2; g++) { trial[nt].code = child_of(b2.nodes[slots[s].code], g); trial[nt].box = box_of(b2.nodes[slots[s].code], g); nt++; } }
                    else trial[nt++] = slots[s];
                }
                if (!fits(trial, nt)) { closed[pick] = true; continue; }
                bool nc[4] = {false, false, false, false};
                for (int s = 0, t = 0; s < ns; s++) { if (s == pick) t += 2; else nc[t++] = closed[s]; }
                for (int s = 0; s < nt; s++) { slots[s] = trial[s]; closed[s] = nc[s]; }
                ns = nt;
            }
        }
        const int32_t idx = (int32_t)out->size();
        out->emplace_back();
        int32_t codes[4];
        for (int s = 0; s < ns; s++) codes[s] = rec(slots[s].code, pending + (ns - 1));
        const float inf = std::numeric_limits<float>::infinity();
        float v[7][4];
        for (int s = 0; s < 4; s++) {
            const bool used = s < ns;
            // an unused slot is a point at +infinity: its entry distance is +inf or its exit -inf, never a hit
            for (int a = 0; a < 3; a++) { v[2 * a][s] = used ? slots[s].box.lo[a] : inf; v[2 * a + 1][s] = used ? slots[s].box.hi[a] : inf; }
            const int32_t code = used ? codes[s] : (int32_t)0x80000000;
            std::memcpy(&v[6][s], &code, 4);
        }
        DNode4& nd = (*out)[idx];
        for (int r = 0; r < 7; r++) nd.q[r] = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
        nd.q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return idx;
    }
};
} // namespace

int32_t collapse_bvh4(const BvhResult& b2, int limit, bool greedy, std::vector<DNode4>* out, int* max_pending) {
    if (b2.nodes.empty() && b2.root >= 0) { *max_pending = 0; return (int32_t)0x80000000; } // defensive: an index without nodes
    std::vector<DNode4> local; // child indices are relative to the first node of this tree, like the BVH2 form
    Collapse c{b2, std::vector<int>(b2.nodes.size(), 0), &local, limit, 0, greedy};
    c.fill_heights(b2.root);
    const int32_t root = c.rec(b2.root, 0);
    out->insert(out->end(), local.begin(), local.end());
    *max_pending = c.max_pending;
    return root;
}

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
    out->root = n ? b.build(0, n, 0) : (int32_t)0x80000000; // no primitives: the "empty" code (RR_SENTINEL), never an index into nodes
    out->order.resize(n);
    for (uint32_t i = 0; i < n; i++) out->order[i] = b.prims[i].id;
    out->depth = b.depth_reached;
    return b.depth_reached <= max_depth;
}

} // namespace rr
