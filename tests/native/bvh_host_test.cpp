// Host-only test of rustray_amd/csrc/rr_bvh.cpp (built with g++ -fsanitize=address,undefined by tests/test_bvh_host.py):
// degenerate inputs of the builder and of the BVH2 -> BVH4 collapse, and the structural invariants the kernels rely on.
#include "../../rustray_amd/csrc/rr_bvh.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <set>

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
static const int32_t SENTINEL = (int32_t)0x80000000;

static int32_t code_of(const float4& q, int k) { const float v[4] = {q.x, q.y, q.z, q.w}; int32_t c; std::memcpy(&c, &v[k], 4); return c; }

// every primitive appears in exactly one leaf of the collapsed tree; returns the deepest pending-stack use
static bool walk4(const std::vector<DNode4>& n4, int32_t code, std::multiset<uint32_t>* seen) {
    if (code == SENTINEL) return true;
    if (code < 0) {
        const uint32_t c = (uint32_t)~code, first = RR_LEAF_FIRST(c), count = RR_LEAF_COUNT(c);
        for (uint32_t i = 0; i < count; i++) seen->insert(first + i);
        return true;
    }
    if ((size_t)code >= n4.size()) return false;
    for (int k = 0; k < 4; k++) if (!walk4(n4, code_of(n4[code].q[6], k), seen)) return false;
    return true;
}

int main() {
    // n = 0: an empty mesh (OBJ group or glTF primitive without faces) and the top level of a zero-item scene
    {
        rr::BvhResult r;
        CHECK(rr::build_bvh(nullptr, nullptr, 0, 8, 24, &r));
        CHECK(r.nodes.empty() && r.order.empty() && r.root == SENTINEL);
        std::vector<DNode4> n4; int pending = -1;
        CHECK(rr::collapse_bvh4(r, 24, true, &n4, &pending) == SENTINEL);
        CHECK(n4.empty() && pending == 0);
        CHECK(rr::collapse_bvh4(r, 12, false, &n4, &pending) == SENTINEL);
    }
    // n = 1: the root is a leaf code
    {
        const float lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
        rr::BvhResult r;
        CHECK(rr::build_bvh(lo, hi, 1, 8, 24, &r));
        CHECK(r.nodes.empty() && r.root < 0 && r.root != SENTINEL && r.order.size() == 1);
        std::vector<DNode4> n4; int pending = -1;
        const int32_t root4 = rr::collapse_bvh4(r, 24, true, &n4, &pending);
        CHECK(root4 == r.root && n4.empty());
    }
    // random soups, identical boxes (no spatial split exists), and a long thin line (deep median splits)
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> u(-10.0f, 10.0f);
    for (int variant = 0; variant < 3; variant++)
        for (uint32_t n : {2u, 9u, 64u, 1000u, 20000u}) {
            std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
            for (uint32_t i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) {
                    float c = variant == 0 ? u(rng) : (variant == 1 ? 1.0f : (k == 0 ? (float)i : 0.0f));
                    lo[3 * (size_t)i + k] = c - 0.1f; hi[3 * (size_t)i + k] = c + 0.1f;
                }
            for (int limit : {24, 12}) {
                for (uint32_t leaf : {1u, 8u}) {
                    rr::BvhResult r;
                    const bool ok = rr::build_bvh(lo.data(), hi.data(), n, leaf, limit, &r);
                    if (!ok) { CHECK((uint64_t)leaf << limit < n); continue; } // only when the depth budget cannot hold n
                    CHECK(r.order.size() == n && r.depth <= limit);
                    std::vector<DNode4> n4; int pending = -1;
                    const int32_t root4 = rr::collapse_bvh4(r, limit, leaf == 8u, &n4, &pending);
                    CHECK(pending <= limit);
                    std::multiset<uint32_t> seen;
                    CHECK(walk4(n4, root4, &seen));
                    CHECK(seen.size() == n && *seen.begin() == 0 && *seen.rbegin() == n - 1 && std::set<uint32_t>(seen.begin(), seen.end()).size() == n);
                }
            }
        }
    // top levels over more items than 2^RR_TLAS_MAX_DEPTH (round 4: rr_scene_create gives them ceil(log2 n) levels, one item per leaf):
    // clustered boxes, so that SAH alone would go deeper than the budget and the median fallback has to hold it
    for (uint32_t n : {4097u, 5000u, 100000u}) {
        std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const float c = (i % 7u == 0u ? 100.0f : 1.0f) * u(rng) + (k == 0 ? 0.001f * (float)i : 0.0f);
                lo[3 * (size_t)i + k] = c - 0.05f; hi[3 * (size_t)i + k] = c + 0.05f;
            }
        int limit = 12;
        while ((1u << limit) < n) limit++;
        rr::BvhResult r;
        CHECK(rr::build_bvh(lo.data(), hi.data(), n, 1, limit, &r));
        CHECK(r.order.size() == n && r.depth <= limit);
        std::vector<DNode4> n4; int pending = -1;
        const int32_t root4 = rr::collapse_bvh4(r, limit, false, &n4, &pending);
        CHECK(pending <= limit);
        std::multiset<uint32_t> seen;
        CHECK(walk4(n4, root4, &seen));
        CHECK(seen.size() == n && std::set<uint32_t>(seen.begin(), seen.end()).size() == n);
        rr::BvhResult small;
        CHECK(!rr::build_bvh(lo.data(), hi.data(), n, 1, limit - 1, &small)); // one level less cannot hold them: what rr_scene_create must never ask for
    }
    std::printf("bvh host test OK\n");
    return 0;
}
