// Bounding-volume hierarchy over the sphere list: the acceleration structure the reference leaves
// as a TODO at the closest-hit loop (main.odin:559).  Built on the host, traversed per lane
// without a stack: nodes are stored in depth-first order, a node whose box the ray hits is
// followed by its first child (the next node), any other by its `skip` link.  Eight copies of the
// order exist, one per sign octant of the ray direction, each with the child on the ray's near
// side first, so the closest hit is usually found early and the far subtrees are culled by it.
#pragma once
#include <cstdint>
#include <vector>

namespace rtk {

constexpr int BVH_LEAF_MAX = 4; // spheres per leaf

// 32 bytes, read as two float4: {bmin.xyz, skip} {bmax.xyz, leaf}
struct BvhNode {
    float bmin[3];
    int32_t skip;  // index of the next node when this one's subtree is not entered; n_nodes = end
    float bmax[3];
    int32_t leaf;  // -1: inner node; otherwise first * 8 + count into the leaf sphere array
};

// 32 bytes, read as two float4: {cx, cy, cz, K} {r*r, 1/r, index, material}
struct BvhSphere {
    float cx, cy, cz, k;   // K = |centre|^2 - r^2, as in the list scan's records
    float r2, inv_r;
    int32_t index;         // position in the caller's sphere list (decides ties, names the hit)
    int32_t material;
};

struct BvhHost {
    int32_t n_nodes = 0;            // nodes per octant
    std::vector<BvhNode> nodes;     // 8 * n_nodes, octant-major; octant bit k set = direction component k negative
    std::vector<BvhSphere> spheres; // leaf order (the same for every octant)
};

// centre_r2: n records {cx, cy, cz, r*r}; k, inv_r, material: n values each
void build_bvh(const float *centre_r2, const float *k, const float *radius, const float *inv_r, const int32_t *material, int32_t n, BvhHost &out);

} // namespace rtk
