// Host-side construction of the sphere BVH (see rt_bvh.h): binned surface-area heuristic over
// the spheres' bounding boxes, leaves of at most BVH_LEAF_MAX spheres, then one depth-first
// linearisation per direction octant.
#include "rt_bvh.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>

namespace rtk {
namespace {

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float *p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double half_area() const {
        const double x = double(hi[0]) - lo[0], y = double(hi[1]) - lo[1], z = double(hi[2]) - lo[2];
        return (x < 0 || y < 0 || z < 0) ? 0.0 : x * y + y * z + z * x;
    }
};

struct TreeNode {
    Box box;
    int left = -1, right = -1; // children (inner node)
    int axis = 0;              // split axis: left holds the smaller coordinates
    int first = 0, count = 0;  // leaf: range of the permutation
};

struct Builder {
    std::vector<Box> boxes;
    std::vector<float> centroid; // 3 per sphere
    std::vector<int> perm;
    std::vector<TreeNode> tree;

    int build(int first, int count) {
        TreeNode node;
        Box cb; // of the centroids
        for (int i = first; i < first + count; ++i) { node.box.grow(boxes[perm[i]]); cb.grow(&centroid[3 * perm[i]]); }
        node.first = first; node.count = count;
        const int id = static_cast<int>(tree.size());
        tree.push_back(node);
        if (count <= 1) return id;

        // binned SAH over the three axes
        constexpr int BINS = 16;
        double best_cost = DBL_MAX; int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            const float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f)) continue;
            Box bb[BINS]; int bc[BINS] = {0};
            for (int i = first; i < first + count; ++i) {
                int b = static_cast<int>(BINS * ((centroid[3 * perm[i] + a] - lo) / ext));
                b = std::min(BINS - 1, std::max(0, b));
                bb[b].grow(boxes[perm[i]]); ++bc[b];
            }
            double right_area[BINS]; int right_count[BINS];
            Box acc; int n = 0;
            for (int b = BINS - 1; b > 0; --b) { acc.grow(bb[b]); n += bc[b]; right_area[b] = acc.half_area(); right_count[b] = n; }
            acc = Box(); n = 0;
            for (int b = 0; b + 1 < BINS; ++b) {
                acc.grow(bb[b]); n += bc[b];
                if (n == 0 || right_count[b + 1] == 0) continue;
                const double cost = acc.half_area() * n + right_area[b + 1] * right_count[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        const double leaf_cost = node.box.half_area() * count;
        int mid = -1, axis = 0;
        // splitting costs one more box test for every ray that reaches this node
        if (best_axis >= 0 && (count > BVH_LEAF_MAX || best_cost + node.box.half_area() < leaf_cost)) {
            axis = best_axis;
            const float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
            auto it = std::partition(perm.begin() + first, perm.begin() + first + count, [&](int s) {
                int b = static_cast<int>(BINS * ((centroid[3 * s + axis] - lo) / ext));
                b = std::min(BINS - 1, std::max(0, b));
                return b <= best_bin;
            });
            mid = static_cast<int>(it - perm.begin());
        } else if (count > BVH_LEAF_MAX) {
            // all centroids coincide (or no split separates them): halve by list position
            mid = first + count / 2;
        }
        if (mid <= first || mid >= first + count) {
            if (count <= BVH_LEAF_MAX) return id;
            mid = first + count / 2;
        }
        const int l = build(first, mid - first);
        const int r = build(mid, first + count - mid);
        tree[id].left = l; tree[id].right = r; tree[id].axis = axis; tree[id].count = 0;
        return id;
    }
};

} // namespace

void build_bvh(const float *centre_r2, const float *k, const float *radius, const float *inv_r, const int32_t *material, int32_t n, BvhHost &out) {
    out = BvhHost();
    if (n <= 0) return;
    auto record = [&](int s) -> BvhSphere {
        const float *c = centre_r2 + 4 * static_cast<size_t>(s);
        return {c[0], c[1], c[2], k[s], c[3], inv_r[s], s, material[s]};
    };
    // spheres more than 16 times the median radius stay outside the hierarchy (the largest BVH_BIG_MAX of them)
    std::vector<float> radii(n);
    for (int i = 0; i < n; ++i) radii[i] = std::fabs(radius[i]);
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float big_radius = 16.0f * sorted[n / 2];
    std::vector<int> big;
    for (int i = 0; i < n; ++i) if (radii[i] > big_radius) big.push_back(i);
    if (static_cast<int>(big.size()) > BVH_BIG_MAX) {
        std::partial_sort(big.begin(), big.begin() + BVH_BIG_MAX, big.end(), [&](int a, int b) { return radii[a] > radii[b] || (radii[a] == radii[b] && a < b); });
        big.resize(BVH_BIG_MAX);
    }
    std::sort(big.begin(), big.end());
    out.n_big = static_cast<int32_t>(big.size());
    for (size_t i = 0; i < big.size(); ++i) out.big[i] = record(big[i]);

    Builder b;
    b.boxes.resize(n); b.centroid.resize(3 * static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        if (std::binary_search(big.begin(), big.end(), i)) continue;
        b.perm.push_back(i);
        const float *c = centre_r2 + 4 * static_cast<size_t>(i);
        const float r = radii[i];
        // The box is wider than the sphere by far more than the rounding of the f32 root
        // evaluation and of the slab test, so every sphere that evaluation can report as hit
        // is reached by the traversal.
        const float pad = std::ldexp(std::max(std::max(std::fabs(c[0]), std::fabs(c[1])), std::fabs(c[2])) + r, -12) + 1e-6f;
        for (int a = 0; a < 3; ++a) { b.boxes[i].lo[a] = c[a] - r - pad; b.boxes[i].hi[a] = c[a] + r + pad; b.centroid[3 * static_cast<size_t>(i) + a] = c[a]; }
    }
    const int m = static_cast<int>(b.perm.size());
    if (m == 0) return;
    b.tree.reserve(2 * static_cast<size_t>(m));
    b.build(0, m);

    // leaf spheres in permutation order, ascending list index inside a leaf
    out.spheres.resize(m);
    for (const TreeNode &t : b.tree)
        if (t.left < 0) std::sort(b.perm.begin() + t.first, b.perm.begin() + t.first + t.count);
    for (int i = 0; i < m; ++i) out.spheres[i] = record(b.perm[i]);

    // quantisation grid over the root box; corners are rounded outwards by one extra step so that
    // the f32 evaluation origin + q*step never lands inside the exact box
    const Box &root = b.tree[0].box;
    double org[3], stp[3];
    for (int a = 0; a < 3; ++a) {
        org[a] = root.lo[a];
        stp[a] = std::max((double(root.hi[a]) - root.lo[a]) / 65533.0, 1e-30);
        out.origin[a] = static_cast<float>(org[a]);
        out.step[a] = static_cast<float>(stp[a]);
        // the grid the kernel evaluates is the rounded f32 one: re-derive it so that quantisation uses the same numbers
        org[a] = out.origin[a]; stp[a] = out.step[a];
    }
    auto quant_lo = [&](float v, int a) { return static_cast<uint16_t>(std::min(65535.0, std::max(0.0, std::floor((double(v) - org[a]) / stp[a]) - 1.0))); };
    auto quant_hi = [&](float v, int a) { return static_cast<uint16_t>(std::min(65535.0, std::max(0.0, std::ceil((double(v) - org[a]) / stp[a]) + 1.0))); };

    const int32_t n_nodes = static_cast<int32_t>(b.tree.size());
    out.n_nodes = n_nodes;
    out.nodes.resize(8 * static_cast<size_t>(n_nodes));
    const bool compact = n_nodes <= BVH_LDS_MAX_NODES && m <= 8192;
    if (compact) {
        out.lds_boxes.resize(n_nodes);
        out.lds_links.assign(8 * static_cast<size_t>(n_nodes), 0u);
        for (int32_t i = 0; i < n_nodes; ++i) {
            const TreeNode &t = b.tree[i];
            uint16_t q[6];
            for (int a = 0; a < 3; ++a) { q[a] = quant_lo(t.box.lo[a], a); q[3 + a] = quant_hi(t.box.hi[a], a); }
            const uint32_t leaf = t.left < 0 ? (0x8000u | (static_cast<uint32_t>(t.count - 1) << 13) | static_cast<uint32_t>(t.first)) : 0u;
            out.lds_boxes[i] = {{q[0] | (uint32_t(q[1]) << 16), q[2] | (uint32_t(q[3]) << 16), q[4] | (uint32_t(q[5]) << 16), leaf}};
        }
    }
    std::vector<int> order; // tree ids in one octant's depth-first order
    for (int oct = 0; oct < 8; ++oct) {
        BvhNode *dst = out.nodes.data() + static_cast<size_t>(oct) * n_nodes;
        order.assign(n_nodes, -1);
        int32_t next = 0;
        // iterative depth-first emission; an inner node's skip link is the index after its subtree
        struct Frame { int node; int32_t slot; int stage; };
        std::vector<Frame> stack;
        stack.push_back({0, -1, 0});
        while (!stack.empty()) {
            Frame &f = stack.back();
            const TreeNode &t = b.tree[f.node];
            if (f.stage == 0) {
                f.slot = next++;
                order[f.slot] = f.node;
                BvhNode &o = dst[f.slot];
                for (int a = 0; a < 3; ++a) { o.lo[a] = quant_lo(t.box.lo[a], a); o.hi[a] = quant_hi(t.box.hi[a], a); }
                if (t.left < 0) {
                    o.link = BVH_LEAF_FLAG | (static_cast<uint32_t>(t.count - 1) << 28) | static_cast<uint32_t>(t.first);
                    stack.pop_back();
                    continue;
                }
                f.stage = 1;
                const bool negative = (oct >> t.axis) & 1; // direction runs from large to small coordinates: right child is nearer
                stack.push_back({negative ? t.right : t.left, -1, 0});
            } else if (f.stage == 1) {
                f.stage = 2;
                const bool negative = (oct >> t.axis) & 1;
                stack.push_back({negative ? t.left : t.right, -1, 0});
            } else {
                dst[f.slot].link = static_cast<uint32_t>(next);
                stack.pop_back();
            }
        }
        if (compact) {
            // the same order expressed on tree ids: slot s holds tree node order[s]; its successors are slot s+1 (hit) and its skip slot (miss)
            uint32_t *links = out.lds_links.data() + static_cast<size_t>(oct) * n_nodes;
            auto id_of = [&](int32_t slot) { return slot >= n_nodes ? BVH_LDS_END : static_cast<uint32_t>(order[slot]); };
            for (int32_t s = 0; s < n_nodes; ++s) {
                const bool is_leaf = (dst[s].link & BVH_LEAF_FLAG) != 0;
                const uint32_t hit = id_of(s + 1), miss = is_leaf ? hit : id_of(static_cast<int32_t>(dst[s].link));
                links[order[s]] = hit | (miss << 16);
            }
        }
    }
}

} // namespace rtk
