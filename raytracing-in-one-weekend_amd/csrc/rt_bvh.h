// Bounding-volume hierarchy over the sphere list: the acceleration structure the reference leaves
// as a TODO at the closest-hit loop (main.odin:559).  Built on the host, traversed per lane
// without a stack: nodes are stored in depth-first order, a node whose box the ray hits is
// followed by its first child (the next node), any other by its `skip` link.  Eight copies of the
// order exist, one per sign octant of the ray direction, each with the child on the ray's near
// side first, so the closest hit is usually found early and the far subtrees are culled by it.
//
// A node is 16 bytes -- one 128-bit load per lane and traversal step, which is what bounds the
// traversal (64 lanes, 64 different cache lines per step): its box is quantised to 16 bits per
// coordinate on a grid over the hierarchy's root box, rounded outwards.  Spheres far larger than
// the typical one (a ground sphere of radius 1000 among spheres of radius 0.2) would make that
// grid coarse; up to BVH_BIG_MAX of them are kept out of the hierarchy and tested for every
// segment instead.
#pragma once
#include <cstdint>
#include <vector>

namespace rtk {

constexpr int BVH_LEAF_MAX = 4; // spheres per leaf
constexpr int BVH_BIG_MAX = 8;  // spheres tested outside the hierarchy
constexpr uint32_t BVH_LEAF_FLAG = 0x80000000u;

// box corner = origin + q * step per axis (q in [0, 65535]).
// link: inner node: index of the next node when this one's subtree is not entered (n_nodes = end);
//       leaf: BVH_LEAF_FLAG | (count - 1) << 28 | first, a range of the leaf sphere array; its
//       successor either way is the next node.
struct BvhNode {
    uint16_t lo[3], hi[3];
    uint32_t link;
};
static_assert(sizeof(BvhNode) == 16, "one 128-bit load per node");

// 32 bytes, read as two float4: {cx, cy, cz, K} {r*r, 1/r, index, material}
struct BvhSphere {
    float cx, cy, cz, k;   // K = |centre|^2 - r^2, as in the list scan's records
    float r2, inv_r;
    int32_t index;         // position in the caller's sphere list (decides ties, names the hit)
    int32_t material;
};

// Compact form for hierarchies small enough to live in LDS (BVH_LDS_MAX_NODES): one copy of the
// boxes in tree order and, per direction octant, two 16-bit links per node.
constexpr int BVH_LDS_MAX_NODES = 1280;      // 1280 * (16 + 8 * 4) bytes = 60 KiB of LDS per workgroup
constexpr uint32_t BVH_LDS_END = 0xffffu;
// {lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, leaf}; leaf = 0 for an inner node,
// otherwise 0x8000 | (count - 1) << 13 | first (a range of the leaf sphere array)
struct BvhBoxLds { uint32_t w[4]; };

struct BvhHost {
    int32_t n_nodes = 0;            // nodes per octant
    std::vector<BvhNode> nodes;     // 8 * n_nodes, octant-major; octant bit k set = direction component k negative
    std::vector<BvhSphere> spheres; // leaf order (the same for every octant)
    float origin[3] = {0, 0, 0}, step[3] = {1, 1, 1}; // the quantisation grid
    // compact form, filled when n_nodes <= BVH_LDS_MAX_NODES and the leaf array has at most 8192 spheres:
    std::vector<BvhBoxLds> lds_boxes;  // n_nodes, tree order (node 0 = root)
    std::vector<uint32_t> lds_links;   // 8 * n_nodes, octant-major: (next node when the box is hit) | (next node when it is missed) << 16
    int32_t n_big = 0;
    BvhSphere big[BVH_BIG_MAX];     // spheres outside the hierarchy, ascending list index
};

// centre_r2: n records {cx, cy, cz, r*r}; k, radius, inv_r, material: n values each
void build_bvh(const float *centre_r2, const float *k /* the filter constant of each sphere, -K' */, const float *radius, const float *inv_r, const int32_t *material, int32_t n, BvhHost &out);

} // namespace rtk
