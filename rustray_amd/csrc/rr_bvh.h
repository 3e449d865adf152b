// rr_bvh.h — host BVH2 builder (see rr_bvh.cpp).
#pragma once
#include <stdint.h>
#include <vector>
#include "rr_device.h"

namespace rr {

struct BvhResult {
    std::vector<DNode> nodes;    // child indices relative to nodes[0]
    std::vector<uint32_t> order; // primitive ids in leaf order
    int32_t root;                // node index, or a leaf code when the whole set is one leaf
    int depth;                   // inner levels on the longest root-to-leaf path
};

// BVH2 -> BVH4: a node adopts its grandchildren wherever a depth-first walk with `limit` stack entries still
// reaches every leaf below (see rr_bvh.cpp).  Appends the nodes to *out (child indices relative to the tree's first
// node), returns the root (node index or leaf code) and the worst-case number of pending entries.
// `greedy`: open the child with the largest box first (per-mesh trees: -4 % on a 320 k-triangle mesh); otherwise both
// children of a node are opened (the top level: greedy measured +2.5 % there).
int32_t collapse_bvh4(const BvhResult& b2, int limit, bool greedy, std::vector<DNode4>* out, int* max_pending);

// boxes_lo / boxes_hi: n * 3 floats.  Returns false if the depth limit could not be met.
bool build_bvh(const float* boxes_lo, const float* boxes_hi, uint32_t n, uint32_t max_leaf, int max_depth, BvhResult* out);

} // namespace rr
