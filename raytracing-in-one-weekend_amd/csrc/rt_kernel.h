// Interface between the C-ABI shim (rt_api.cpp) and the gfx950 kernels (rt_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtk {

constexpr int TILE = 8;              // a wave64 owns one 8x8-pixel tile, lane l -> (l & 7, l >> 3)
constexpr int WAVES_PER_BLOCK = 4;   // 256-thread workgroups: four tiles share one LDS copy of the spheres
constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;
constexpr int BVH_LDS_BLOCK_THREADS = 512; // hierarchy in LDS: eight waves share one copy
constexpr int GROUP = 4;             // sphere records are fetched and tested four at a time
constexpr int SCAN_STEP = 2 * GROUP;  // spheres per iteration of the scan loop (two register sets)
constexpr int WQ_CAP = 256;          // candidates a wave queues in LDS before it evaluates them (more are evaluated early)
constexpr int REJECTION_CAP = 64;    // bound on the two rejection-sampling loops (P(reaching it) < 1e-20)
constexpr int LDS_MAX_SPHERES_F32 = 4088; // just under 64 KiB of float4 records per workgroup
constexpr int LDS_MAX_SPHERES_F64 = 2040; // just under 64 KiB of double4 records per workgroup

// ARITH_F32_FMA    : f32, fused scan, work queue, accumulators are Q32.32 in uint64 (atomic adds)
// ARITH_F64        : f64, reference operation order, accumulators are double
// ARITH_F32_LITERAL: f32, reference operation order, accumulators are float
// ARITH_MIXED      : f64, reference operation order, behind the packed-f32 scan used as a conservative
//                    filter; work queue; every sample colour is staged and summed in sample order
// ARITH_MIXED_FIXED: the paths of ARITH_MIXED; every sample colour is added to its pixel's accumulators as a Q32.64
//                    fixed-point number, two uint64 words per channel {units of 2^-32, units of 2^-64} (atomic adds)
enum Arith { ARITH_F32_FMA = 0, ARITH_F64 = 1, ARITH_F32_LITERAL = 2, ARITH_MIXED = 3, ARITH_MIXED_FIXED = 4 };
inline size_t accum_element_size(int arith) { return arith == ARITH_F32_LITERAL ? 4 : (arith == ARITH_MIXED_FIXED ? 16 : 8); }
enum Fetch { FETCH_SCALAR = 1, FETCH_LDS = 2, FETCH_BVH = 3 };

// camera_render_data (main.odin:440-448) as 19 numbers in the kernel's arithmetic type:
// position, pixel_delta_u, pixel_delta_v, view_plane_top_left_pixel_center, dof_disk_u,
// dof_disk_v (3 each) and depth_of_field_angle.
struct CameraF32 { float v[19]; };
struct CameraF64 { double v[19]; };

// Which 8-row tile rows of the region a launch owns: tile row k of the launch is tile row
// ((k / rows_per_strip) * strip_count + strip_index) * rows_per_strip + k % rows_per_strip
// of the region.
struct TileMap {
    int32_t tiles_x;        // tiles per tile row
    int32_t n_tiles;        // tiles owned by this launch
    int32_t rows_per_strip; // tile rows per strip
    int32_t strip_index, strip_count;
};

struct RenderParams {
    // world, structure of arrays (device pointers)
    const void *sphere_rec;     // scan records: float4 / double4 {cx, cy, cz, r*r}; throughput and mixed mode: per PAIR
                                // of spheres {cx0,cx1, cy0,cy1, cz0,cz1, -K'0,-K'1}, K' = |c|^2 - r*r lowered by the
                                // filter's margin 2^-16 (|c|^2 + r*r)
    float filter_reach;         // filter modes: max over spheres of |c| + |r|, rounded up; FLT_MAX for a world outside the
                                // range of the filter's error bound (every ray then keeps every sphere)
    const void *sphere_center;  // float4 / double4 {cx, cy, cz, r*r}: the centre is read from here on a hit
    const void *sphere_radius;  // per sphere radius, float / double (throughput mode: 1/radius)
    const int32_t *sphere_material;
    const void *material_rec;   // per material {albedo.xyz, param1} in float4 / double4
    const int32_t *material_type;
    int32_t n_spheres;          // padded to a multiple of SCAN_STEP (pad records have r*r = -inf); GROUP more follow
    // FETCH_BVH: the hierarchy (rt_bvh.h): 8 octant orders of bvh_n_nodes 16-byte nodes, leaf spheres (two float4 each),
    // the quantisation grid of the node boxes, and the spheres kept outside the hierarchy
    const void *bvh_nodes;
    const void *bvh_spheres;
    int32_t bvh_n_nodes;
    int32_t bvh_n_big;
    const void *bvh_lds_boxes;  // compact form staged into LDS by every workgroup (NULL: the hierarchy is too large for that)
    const void *bvh_lds_links;
    float bvh_origin[3], bvh_step[3];
    float bvh_big[8][8];        // {cx, cy, cz, K, r*r, 1/r, index, material} (the last two as integer bits)
    // image and work
    int32_t width, height;
    int32_t min_x, min_y, max_x, max_y; // region, half open
    TileMap map;
    uint32_t sample_begin;   // first global sample index of the call
    uint32_t sample_count;   // samples of the call
    uint32_t chunk_samples;  // samples per chunk
    uint32_t n_items;        // queue mode: n_tiles * chunks work items, item = chunk * n_tiles + tile
    uint32_t *queue;         // queue mode: head of the work queue (zeroed before the launch)
    int32_t max_bounces;
    uint32_t k0, k1;         // RNG seed keys
    void *accum;             // width*height*3 sums (element type: see Arith); literal modes: chunk 0 writes here
    void *slabs;             // chunk c >= 1 writes to slabs + (c-1)*slab_stride (same layout); mixed mode: the
                             // staged sample colours, sample s of the launch at slabs + s*slab_stride
    int64_t slab_stride;     // elements per slab (mixed mode: owned tiles * 64 * 3)
    unsigned long long *stats; // optional device counters
    CameraF32 cam32;
    CameraF64 cam64;
};

// resident_blocks: workgroups the device holds at once (queue mode launches no more than that)
hipError_t launch_render(const RenderParams &p, int arith, int fetch, int chunks, int resident_blocks, hipStream_t stream);

struct ReduceParams {
    void *accum; const void *slabs; int64_t slab_stride; int32_t chunks;
    int32_t words_per_pixel; // launch_clear: 64-bit words of accumulator per pixel (3: Q32.32, 6: Q32.64)
    int32_t width; int32_t min_x, min_y, max_x, max_y; TileMap map;
};
// accum[p] = accum[p] + slab_0[p] + slab_1[p] + ... (chunk order) for the pixels the TileMap owns
hipError_t launch_reduce(const ReduceParams &p, int arith, hipStream_t stream);
// queue mode: zero the owned pixels' accumulators and the queue head
hipError_t launch_clear(const ReduceParams &p, uint32_t *queue, hipStream_t stream);
// mixed mode: zero the queue head only
hipError_t launch_clear_queue(uint32_t *queue, hipStream_t stream);
// mixed mode: accum[p] = (first ? 0 : accum[p]) + staged_0[p] + staged_1[p] + ... for p.chunks staged samples, in order
hipError_t launch_order_sum(const ReduceParams &p, bool first, hipStream_t stream);

struct FinalizeParams {
    const void *accum; double *image; double *linear; // image/linear: 3 doubles per pixel
    int64_t total_samples;
    int32_t width; int32_t min_x, min_y, max_x, max_y; TileMap map;
};
hipError_t launch_finalize(const FinalizeParams &p, int arith, hipStream_t stream);

// The same tail followed by the body of serialize_ppm (main.odin:806-820): 12 bytes of text per
// pixel, "RRR GGG BBB\n", written as three 32-bit words at text[3 * (u + v*width)].
struct PpmParams {
    const void *accum; uint32_t *text;
    int64_t total_samples;
    int32_t width; int32_t min_x, min_y, max_x, max_y; TileMap map;
};
hipError_t launch_finalize_ppm(const PpmParams &p, int arith, hipStream_t stream);

// Scratch for callers that cannot allocate: one file-scope device array per GPU, held until the
// process ends.  Its first STATIC_SCRATCH_HEAD_BYTES hold the work-queue head, the rest chunk slabs
// or staged sample colours.  static_scratch_address returns the array's address on `device`, which
// must be the current device the first time it is asked for (the lookup is cached per device).
constexpr size_t STATIC_SCRATCH_BYTES = static_cast<size_t>(32) << 20;
constexpr size_t STATIC_SCRATCH_HEAD_BYTES = 256;
hipError_t static_scratch_address(int device, void **out);

} // namespace rtk
