// C ABI of the library (include/rt_mi355x.h): argument validation, AoS f64 -> SoA upload,
// launch bookkeeping and the host <-> device copies of the one-shot entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_bvh.h"
#include "rt_kernel.h"

namespace {

thread_local std::string g_error = "";

int32_t fail(int32_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// One lock per device ordinal: the one-shot entry points may be called from several host
// threads (the reference calls render_region from pool threads, main.odin:609).  Ordinals are
// validated against MAX_DEVICES before any table indexed by them is touched.
constexpr int MAX_DEVICES = 64;
std::mutex &device_mutex(int device) {
    static std::mutex m[MAX_DEVICES];
    return m[device];
}

uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
void seed_keys(uint64_t seed, uint32_t &k0, uint32_t &k1) {
    k0 = mix32(static_cast<uint32_t>(seed) ^ 0x9e3779b9u);
    k1 = mix32(static_cast<uint32_t>(seed >> 32) ^ k0 ^ 0x85ebca6bu);
}

bool precision_ok(int32_t p) { return p == RT_PRECISION_F32 || p == RT_PRECISION_F64 || p == RT_PRECISION_F32_LITERAL || p == RT_PRECISION_MIXED || p == RT_PRECISION_MIXED_FIXED; }
static_assert(RT_PRECISION_F32 == rtk::ARITH_F32_FMA && RT_PRECISION_F64 == rtk::ARITH_F64 && RT_PRECISION_F32_LITERAL == rtk::ARITH_F32_LITERAL &&
              RT_PRECISION_MIXED == rtk::ARITH_MIXED && RT_PRECISION_MIXED_FIXED == rtk::ARITH_MIXED_FIXED, "the ABI's precision values are the kernels' arithmetic modes");
size_t accum_size(int32_t precision) { return rtk::accum_element_size(precision); }

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (device >= 0 && device != prev) {
            e = hipSetDevice(device);
            if (e != hipSuccess) return e;
            active = true;
        }
        return hipSuccess;
    }
    ~DeviceGuard() { if (active) (void)hipSetDevice(prev); }
};

} // namespace

// What one rt_render_accumulate call owns while it is in flight: a work-queue head and the two
// events that time it.  A world holds a small ring of them, so calls on different streams do not
// share a queue head.
struct CallSlot {
    uint32_t *queue = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool used = false;
};
constexpr int CALL_SLOTS = 8;

struct rt_world_s {
    int device = 0;
    int32_t n_spheres = 0, n_padded = 0, n_materials = 0;
    int resident_blocks = 2048;                 // workgroups the device can hold at once
    uint32_t *queue_ring = nullptr;             // CALL_SLOTS work-queue heads, 64 bytes apart (a call whose windows overlap uses the word 32 bytes on as well)
    CallSlot slots[CALL_SLOTS];
    uint64_t n_calls = 0;
    int last_slot = -1;
    void *rec32 = nullptr, *rec64 = nullptr;   // {cx, cy, cz, r*r}
    void *rec32m = nullptr;                     // the filter's records, per pair of spheres {cx0,cx1, cy0,cy1, cz0,cz1, -K'0,-K'1}: K = |c|^2 - r*r lowered by the filter margin
    float filter_reach = 0.0f;                  // max |c| + |r| over the spheres, rounded up
    bool filter_ok = true;                      // every sphere inside the range the filter's error bound needs
    void *rad32 = nullptr, *rad64 = nullptr;
    void *rad32inv = nullptr;                   // 1/radius in f32, the throughput kernel's hit normal is (p - c) * (1/r)
    int32_t *sphere_material = nullptr;
    void *mat32 = nullptr, *mat64 = nullptr;   // {albedo, param1}
    int32_t *material_type = nullptr;
    // RT_FETCH_BVH: built and uploaded on first use from the host copies kept here
    std::vector<float> h_rec32, h_km, h_rad32, h_rad32inv; // h_km: -K' of the filter (K lowered by its margin, negated)
    std::vector<int32_t> h_material;
    void *bvh_nodes = nullptr, *bvh_spheres = nullptr, *bvh_lds_boxes = nullptr, *bvh_lds_links = nullptr;
    int32_t bvh_n_nodes = 0;
    bool bvh_built = false;
    rtk::BvhHost bvh_host;                      // grid and big-sphere list (node and sphere vectors are released after upload)
    // Scratch the kernels of a call write and read: partial sums of sample chunks 1.. (literal modes)
    // or the staged sample colours (mixed mode).  One buffer per world: a call that uses it on
    // another stream than the previous such call first waits (on the device) for that call's end.
    void *slabs = nullptr;
    size_t slab_bytes = 0;
    size_t staging_cap = ~static_cast<size_t>(0); // half the smallest staging request the device has refused so far
    hipEvent_t slabs_done = nullptr;
    hipStream_t slabs_stream = nullptr;
    bool slabs_busy = false;
    // Ordered mixed mode with more samples than the staging buffer holds: the windows alternate between the halves of the
    // buffer and between two streams of the world's own, so that a window's first waves fill the compute units its
    // predecessor's last waves leave idle (a window on its own costs about a millisecond of ramp-up and tail).
    hipStream_t win_stream[2] = {nullptr, nullptr};
    hipEvent_t win_fork = nullptr, win_sum[2] = {nullptr, nullptr}, win_join[2] = {nullptr, nullptr};
    void *static_scratch = nullptr;             // this device's copy of the kernels' static scratch array
    std::mutex mu;
};

namespace {

int32_t validate_scene(const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials) {
    if (n_spheres < 0 || n_materials < 0) return fail(RT_ERR_INVALID_ARGUMENT, "negative element count (%lld spheres, %lld materials)", (long long)n_spheres, (long long)n_materials);
    if (n_spheres > 0 && spheres == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "spheres is NULL");
    if (n_materials > 0 && materials == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "materials is NULL");
    if (n_spheres > (1 << 24) || n_materials > (1 << 24)) return fail(RT_ERR_UNSUPPORTED, "more than 2^24 spheres or materials");
    for (int64_t i = 0; i < n_materials; ++i) {
        const rt_material &m = materials[i];
        if (m.type < RT_MATERIAL_LAMBERTIAN || m.type > RT_MATERIAL_METALLIC_HOFFMAN_PAPER) return fail(RT_ERR_INVALID_SCENE, "material %lld has unknown type %lld", (long long)i, (long long)m.type);
        // make_metallic_data asserts 0 <= fuzz <= 1 (main.odin:280)
        if ((m.type == RT_MATERIAL_METALLIC || m.type >= RT_MATERIAL_METALLIC_FRESNEL) && !(0.0 <= m.param1 && m.param1 <= 1.0)) return fail(RT_ERR_INVALID_SCENE, "metallic material %lld has fuzz %g outside [0, 1]", (long long)i, m.param1);
        if (m.type == RT_MATERIAL_DIELECTRIC && !(m.param1 > 0.0) ) return fail(RT_ERR_INVALID_SCENE, "dielectric material %lld has refractive index %g <= 0", (long long)i, m.param1);
    }
    for (int64_t i = 0; i < n_spheres; ++i) {
        const rt_sphere &s = spheres[i];
        if (s.material_index < 0 || s.material_index >= n_materials) return fail(RT_ERR_INVALID_SCENE, "sphere %lld has material_index %lld outside [0, %lld)", (long long)i, (long long)s.material_index, (long long)n_materials);
        if (!(s.radius != 0.0) || !std::isfinite(s.radius)) return fail(RT_ERR_INVALID_SCENE, "sphere %lld has radius %g", (long long)i, s.radius);
        if (!std::isfinite(s.center.x) || !std::isfinite(s.center.y) || !std::isfinite(s.center.z)) return fail(RT_ERR_INVALID_SCENE, "sphere %lld has a non-finite centre", (long long)i);
    }
    return RT_OK;
}

struct Work { // validated geometry of one call
    rtk::TileMap map;
    int32_t min_x, min_y, max_x, max_y;
};

int32_t plan_work(int64_t width, int64_t height, rt_rect region, int64_t strip_rows, int32_t strip_index, int32_t strip_count, Work &w) {
    if (width <= 0 || height <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "image size %lld x %lld", (long long)width, (long long)height);
    if (width * height > (int64_t(1) << 30)) return fail(RT_ERR_UNSUPPORTED, "image of %lld pixels exceeds 2^30", (long long)(width * height));
    if (region.min_x < 0 || region.min_y < 0 || region.one_past_max_x > width || region.one_past_max_y > height ||
        region.min_x > region.one_past_max_x || region.min_y > region.one_past_max_y)
        return fail(RT_ERR_INVALID_ARGUMENT, "region {%lld, %lld, %lld, %lld} is not inside the %lld x %lld image", (long long)region.min_x, (long long)region.min_y,
                    (long long)region.one_past_max_x, (long long)region.one_past_max_y, (long long)width, (long long)height);
    if (strip_rows <= 0 || strip_rows % rtk::TILE != 0) return fail(RT_ERR_INVALID_ARGUMENT, "strip_rows %lld is not a positive multiple of 8", (long long)strip_rows);
    if (strip_count < 1 || strip_index < 0 || strip_index >= strip_count) return fail(RT_ERR_INVALID_ARGUMENT, "strip %d of %d", strip_index, strip_count);
    const int64_t rw = region.one_past_max_x - region.min_x, rh = region.one_past_max_y - region.min_y;
    const int64_t tiles_x = (rw + rtk::TILE - 1) / rtk::TILE;
    const int64_t tile_rows = (rh + rtk::TILE - 1) / rtk::TILE;
    const int64_t rps = strip_rows / rtk::TILE;
    const int64_t strips = (tile_rows + rps - 1) / rps;
    int64_t owned_rows = 0;
    for (int64_t q = strip_index; q < strips; q += strip_count) owned_rows += std::min<int64_t>(rps, tile_rows - q * rps);
    w.map.tiles_x = static_cast<int32_t>(tiles_x);
    w.map.n_tiles = static_cast<int32_t>(tiles_x * owned_rows);
    w.map.rows_per_strip = static_cast<int32_t>(rps);
    w.map.strip_index = strip_index;
    w.map.strip_count = strip_count;
    w.min_x = static_cast<int32_t>(region.min_x); w.min_y = static_cast<int32_t>(region.min_y);
    w.max_x = static_cast<int32_t>(region.one_past_max_x); w.max_y = static_cast<int32_t>(region.one_past_max_y);
    return RT_OK;
}

void pack_camera(const rt_camera_render_data *cam, rtk::CameraF32 &c32, rtk::CameraF64 &c64) {
    const rt_v3 *v[6] = {&cam->position, &cam->pixel_delta_u, &cam->pixel_delta_v, &cam->view_plane_top_left_pixel_center, &cam->dof_disk_u, &cam->dof_disk_v};
    for (int i = 0; i < 6; ++i) {
        c64.v[3 * i] = v[i]->x; c64.v[3 * i + 1] = v[i]->y; c64.v[3 * i + 2] = v[i]->z;
    }
    c64.v[18] = cam->depth_of_field_angle;
    for (int i = 0; i < 19; ++i) c32.v[i] = static_cast<float>(c64.v[i]);
}

int auto_chunks(int32_t n_tiles, int64_t sample_count) {
    // enough waves that the last ones to finish are short next to the whole launch
    const int64_t target_waves = 32768;
    int64_t c = (target_waves + n_tiles - 1) / std::max(1, n_tiles);
    c = std::min<int64_t>(c, std::max<int64_t>(1, sample_count / 16));
    return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(c, 32)));
}

// Samples per pixel of a work item of the queue modes: 4 (256 paths per item); 8 for the fixed-point mixed mode, whose
// tile sums cost six global atomics per pixel and item (measured on the headline frame: 4 -> 8 samples -0.9 %, 16 +0.2 %;
// the other queue modes are flat from 4 to 8 and lose 1-2 % at 16).  RT_MI355X_UNIT_SAMPLES overrides it (a tuning aid).
int64_t unit_samples(bool fixed_point_mixed = false) {
    static const int64_t override_value = [] {
        const char *env = std::getenv("RT_MI355X_UNIT_SAMPLES");
        const long long v = (env != nullptr && env[0] != '\0') ? std::atoll(env) : 0;
        return static_cast<int64_t>(v >= 1 && v <= 4096 ? v : 0);
    }();
    return override_value != 0 ? override_value : (fixed_point_mixed ? 8 : 4);
}

int auto_queue_chunks(int64_t sample_count, bool fixed_point_mixed) {
    // work items of a few samples per pixel (some hundred paths): small enough that the items still
    // running when the queue empties are a sliver of the launch, large enough that taking one
    // (an atomic and a few scalar instructions per wave) and flushing a lane's sums (a set of atomics
    // per unit) cost nothing
    const int64_t unit = unit_samples(fixed_point_mixed);
    return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((sample_count + unit - 1) / unit, 4096)));
}

std::atomic<uint64_t> g_call_counter{0};

} // namespace

extern "C" {

const char *rt_version(void) { return "rt_mi355x 0.1 (gfx950)"; }
const char *rt_last_error(void) { return g_error.c_str(); }

int32_t rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

void rt_options_default(rt_options *o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->rng_mode = RT_RNG_DETERMINISTIC;
    o->precision = RT_PRECISION_MIXED; // the reference computes in f64 (main.odin:19): the default returns that image
    o->sphere_fetch = RT_FETCH_AUTO;
    o->device = -1;
    o->device_count = 1;
}

int32_t rt_world_create(const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials, int32_t device, rt_world **out) {
    if (out == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int32_t rc = validate_scene(spheres, n_spheres, materials, n_materials);
    if (rc != RT_OK) return rc;
    DeviceGuard guard;
    HIP_TRY(guard.enter(device));
    int current = 0;
    HIP_TRY(hipGetDevice(&current));

    const size_t ns = static_cast<size_t>(n_spheres), nm = static_cast<size_t>(n_materials);
    // records are padded to a multiple of SCAN_STEP, plus one group the scan may prefetch; a pad
    // record has r*r = -inf, which makes its discriminant -inf (or NaN): never a candidate
    const size_t ns_pad = (ns + rtk::SCAN_STEP - 1) / rtk::SCAN_STEP * rtk::SCAN_STEP;
    const size_t ns_alloc = ns_pad + rtk::GROUP;
    std::vector<float> rec32(4 * ns_alloc, 0.0f), rec32m(4 * ns_alloc, 0.0f), rad32(ns), rad32inv(ns), mat32(4 * nm);
    std::vector<double> rec64(4 * ns_alloc, 0.0), rad64(ns), mat64(4 * nm);
    // the filter's records are stored per pair of spheres: {cx0,cx1, cy0,cy1, cz0,cz1, -K'0,-K'1}
    auto pair_slot = [](size_t i, int component) { return 8 * (i / 2) + 2 * component + (i % 2); };
    for (size_t i = ns; i < ns_alloc; ++i) { rec32[4 * i + 3] = -INFINITY; rec64[4 * i + 3] = -INFINITY; rec32m[pair_slot(i, 3)] = -3.402823466e+38f; }
    double reach = 0.0;
    bool filter_ok = true;
    std::vector<int32_t> smat(ns), mtype(nm);
    for (size_t i = 0; i < ns; ++i) {
        const rt_sphere &s = spheres[i];
        const float r = static_cast<float>(s.radius);
        rec32[4 * i + 0] = static_cast<float>(s.center.x); rec32[4 * i + 1] = static_cast<float>(s.center.y);
        rec32[4 * i + 2] = static_cast<float>(s.center.z); rec32[4 * i + 3] = r * r; // sphere_radius*sphere_radius, main.odin:248
        // K = |centre|^2 - r^2 of the f32 centre and radius, formed in f64
        const double c2 = static_cast<double>(rec32[4 * i]) * rec32[4 * i] + static_cast<double>(rec32[4 * i + 1]) * rec32[4 * i + 1] + static_cast<double>(rec32[4 * i + 2]) * rec32[4 * i + 2];
        const double r2 = static_cast<double>(r) * r;
        // the filter (throughput and mixed mode) adds -K' to the square of c.e, K' = K lowered by the sphere's share of the filter
        // margin, 2^-16 (|c|^2 + r^2); the record holds -K' rounded up (K' rounded down: the side that keeps more)
        {
            const double lowered = (c2 - r2) - (c2 + r2) * (1.0 / 65536.0);
            float km = static_cast<float>(-lowered);
            if (static_cast<double>(km) < -lowered) km = std::nextafterf(km, INFINITY);
            for (int k = 0; k < 3; ++k) rec32m[pair_slot(i, k)] = rec32[4 * i + k];
            rec32m[pair_slot(i, 3)] = km;
            const double far = std::sqrt(c2) + std::fabs(static_cast<double>(r));
            reach = std::max(reach, far);
            // the filter's error bound holds while every product stays a normal f32 number
            if (!(std::fabs(static_cast<double>(r)) >= 1.0 / 1048576.0) || !(far <= 1048576.0)) filter_ok = false;
        }
        rec64[4 * i + 0] = s.center.x; rec64[4 * i + 1] = s.center.y; rec64[4 * i + 2] = s.center.z; rec64[4 * i + 3] = s.radius * s.radius;
        rad32[i] = r; rad64[i] = s.radius; rad32inv[i] = 1.0f / r;
        smat[i] = static_cast<int32_t>(s.material_index);
    }
    for (size_t i = 0; i < nm; ++i) {
        const rt_material &m = materials[i];
        mat64[4 * i + 0] = m.albedo.x; mat64[4 * i + 1] = m.albedo.y; mat64[4 * i + 2] = m.albedo.z; mat64[4 * i + 3] = m.param1;
        for (int k = 0; k < 4; ++k) mat32[4 * i + k] = static_cast<float>(mat64[4 * i + k]);
        mtype[i] = static_cast<int32_t>(m.type);
    }
    rt_world *w = new rt_world_s();
    w->h_rec32.assign(rec32.begin(), rec32.begin() + 4 * ns);
    w->h_km.resize(ns);
    for (size_t i = 0; i < ns; ++i) w->h_km[i] = rec32m[pair_slot(i, 3)];
    w->h_rad32 = rad32; w->h_rad32inv = rad32inv; w->h_material = smat;
    w->filter_ok = filter_ok;
    w->filter_reach = static_cast<float>(reach);
    if (static_cast<double>(w->filter_reach) < reach) w->filter_reach = std::nextafterf(w->filter_reach, INFINITY);
    w->device = current;
    w->n_spheres = static_cast<int32_t>(n_spheres);
    w->n_padded = static_cast<int32_t>(ns_pad);
    w->n_materials = static_cast<int32_t>(n_materials);
    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 64));
        if (e != hipSuccess) return e;
        if (bytes == 0) return hipSuccess;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = upload(&w->rec32, rec32.data(), rec32.size() * 4);
    if (e == hipSuccess) e = upload(&w->rec32m, rec32m.data(), rec32m.size() * 4);
    if (e == hipSuccess) e = upload(&w->rec64, rec64.data(), rec64.size() * 8);
    if (e == hipSuccess) e = upload(&w->rad32, rad32.data(), rad32.size() * 4);
    if (e == hipSuccess) e = upload(&w->rad64, rad64.data(), rad64.size() * 8);
    if (e == hipSuccess) e = upload(&w->rad32inv, rad32inv.data(), rad32inv.size() * 4);
    if (e == hipSuccess) e = upload(reinterpret_cast<void **>(&w->sphere_material), smat.data(), smat.size() * 4);
    if (e == hipSuccess) e = upload(&w->mat32, mat32.data(), mat32.size() * 4);
    if (e == hipSuccess) e = upload(&w->mat64, mat64.data(), mat64.size() * 8);
    if (e == hipSuccess) e = upload(reinterpret_cast<void **>(&w->material_type), mtype.data(), mtype.size() * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&w->queue_ring), 64 * CALL_SLOTS);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, current);
        // 2048 threads per CU = 8 workgroups of 256; a grid larger than what is resident is harmless
        // (late workgroups find the queue empty), a smaller one would leave CUs idle
        if (e == hipSuccess) w->resident_blocks = std::max(1, prop.multiProcessorCount) * (2048 / rtk::BLOCK_THREADS);
        // RT_MI355X_TEST_RESIDENT_BLOCKS (a testing aid): a grid of that many workgroups, so that a small image gives
        // every wave many work items (the tile sums' generations and late samples, the queue's refills)
        if (const char *t = std::getenv("RT_MI355X_TEST_RESIDENT_BLOCKS")) {
            const long v = std::strtol(t, nullptr, 10);
            if (v >= 1 && v <= 1 << 20) w->resident_blocks = static_cast<int>(v);
        }
    }
    for (int k = 0; k < CALL_SLOTS && e == hipSuccess; ++k) {
        w->slots[k].queue = w->queue_ring + 16 * k;
        e = hipEventCreate(&w->slots[k].ev_begin);
        if (e == hipSuccess) e = hipEventCreate(&w->slots[k].ev_end);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w->slabs_done, hipEventDisableTiming);
    if (e == hipSuccess) e = rtk::static_scratch_address(current, &w->static_scratch); // set-up for rt_render_accumulate_static
    if (e != hipSuccess) {
        rt_world_destroy(w);
        return fail(RT_ERR_DEVICE, "uploading the world failed: %s", hipGetErrorString(e));
    }
    *out = w;
    return RT_OK;
}

int32_t rt_world_destroy(rt_world *w) {
    if (w == nullptr) return RT_OK;
    DeviceGuard guard;
    (void)guard.enter(w->device);
    // calls still in flight read the buffers freed below
    for (CallSlot &c : w->slots) if (c.used && c.ev_end) (void)hipEventSynchronize(c.ev_end);
    void *bufs[] = {w->rec32, w->rec32m, w->rec64, w->rad32, w->rad32inv, w->rad64, w->sphere_material, w->mat32, w->mat64, w->material_type, w->slabs, w->queue_ring, w->bvh_nodes, w->bvh_spheres, w->bvh_lds_boxes, w->bvh_lds_links};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (CallSlot &c : w->slots) {
        if (c.ev_begin) (void)hipEventDestroy(c.ev_begin);
        if (c.ev_end) (void)hipEventDestroy(c.ev_end);
    }
    if (w->slabs_done) (void)hipEventDestroy(w->slabs_done);
    for (int k = 0; k < 2; ++k) {
        if (w->win_stream[k]) { (void)hipStreamSynchronize(w->win_stream[k]); (void)hipStreamDestroy(w->win_stream[k]); }
        if (w->win_sum[k]) (void)hipEventDestroy(w->win_sum[k]);
        if (w->win_join[k]) (void)hipEventDestroy(w->win_join[k]);
    }
    if (w->win_fork) (void)hipEventDestroy(w->win_fork);
    delete w;
    return RT_OK;
}

} // extern "C"

namespace {

// Where a call's device scratch comes from.  The regular entry point takes the queue head from the
// world's ring and lets the world's scratch buffer grow (hipMalloc); the entry point that allocates
// nothing brings one fixed region per device and fails when the call does not fit.
struct ScratchSource {
    uint32_t *queue = nullptr;   // NULL: use the call slot's
    void *work = nullptr;        // NULL: the world's buffer, grown on demand
    size_t work_bytes = 0;
    bool fixed = false;
};

size_t env_mib(const char *name, size_t fallback) {
    const char *env = std::getenv(name);
    if (env != nullptr && env[0] != '\0') {
        const long long mib = std::atoll(env);
        if (mib > 0) return static_cast<size_t>(mib) << 20;
    }
    return fallback;
}
// default upper bound of the ordered mixed mode's staging buffer (it holds 24 bytes per pixel and staged sample;
// a call whose samples do not fit is run as several launches, each of which costs about a millisecond of ramp-up
// and tail); RT_MI355X_STAGING_MIB overrides it.  An allocation that fails is retried at half the size.
size_t staging_budget() { return env_mib("RT_MI355X_STAGING_MIB", static_cast<size_t>(4) << 30); }
// hipMalloc for the staging buffer.  RT_MI355X_TEST_MALLOC_LIMIT_MIB (a testing aid) makes requests above that size
// fail as if the device were out of memory, so that the retry path can be exercised on a 288 GB device.
hipError_t staging_malloc(void **ptr, size_t bytes) {
    const size_t limit = env_mib("RT_MI355X_TEST_MALLOC_LIMIT_MIB", ~static_cast<size_t>(0));
    if (bytes > limit) return hipErrorOutOfMemory;
    const hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) { *ptr = nullptr; (void)hipGetLastError(); }
    return e;
}

int32_t accumulate_impl(rt_world *world, const rt_camera_render_data *cam, int64_t width, int64_t height, rt_rect region,
                        int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                        int64_t sample_begin, int64_t sample_count, uint64_t seed, int32_t precision, int32_t sphere_fetch,
                        int32_t sample_chunks, void *d_accum, uint64_t *d_stats, void *stream, ScratchSource src) {
    if (world == nullptr || cam == nullptr || d_accum == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "world, cam or d_accum is NULL");
    if (!precision_ok(precision)) return fail(RT_ERR_INVALID_ARGUMENT, "unknown precision %d", precision);
    if (sphere_fetch < RT_FETCH_AUTO || sphere_fetch > RT_FETCH_BVH_GLOBAL) return fail(RT_ERR_INVALID_ARGUMENT, "unknown sphere_fetch %d", sphere_fetch);
    const bool bvh_global_only = sphere_fetch == RT_FETCH_BVH_GLOBAL;
    if (bvh_global_only) sphere_fetch = RT_FETCH_BVH;
    if (sphere_fetch == RT_FETCH_BVH && precision != RT_PRECISION_F32 && precision != RT_PRECISION_MIXED && precision != RT_PRECISION_MIXED_FIXED)
        return fail(RT_ERR_UNSUPPORTED, "RT_FETCH_BVH is available with RT_PRECISION_F32, RT_PRECISION_MIXED and RT_PRECISION_MIXED_FIXED only");
    if (sample_begin < 0 || sample_count < 0 || sample_begin + sample_count > (int64_t(1) << 32) - 2) return fail(RT_ERR_INVALID_ARGUMENT, "samples [%lld, +%lld) outside [0, 2^32-2)", (long long)sample_begin, (long long)sample_count);
    if (cam->max_ray_bounces < 0 || cam->max_ray_bounces > (1 << 20)) return fail(RT_ERR_INVALID_ARGUMENT, "max_ray_bounces %lld outside [0, 2^20]", (long long)cam->max_ray_bounces);
    if (sample_chunks < 0) return fail(RT_ERR_INVALID_ARGUMENT, "sample_chunks %d < 0", sample_chunks);
    Work wk;
    int32_t rc = plan_work(width, height, region, strip_rows, strip_index, strip_count, wk);
    if (rc != RT_OK) return rc;
    if (wk.map.n_tiles == 0) return RT_OK;

    // The mixed mode's filter needs a world inside its range; any other world is traced by the f64
    // kernel with one sample chunk, which is the same arithmetic and the same summation order.
    bool mixed = precision == RT_PRECISION_MIXED;
    if (mixed && !world->filter_ok) { mixed = false; precision = RT_PRECISION_F64; sample_chunks = 1; if (sphere_fetch == RT_FETCH_BVH) sphere_fetch = RT_FETCH_SCALAR; }
    // fixed-point sums: the same kernel, and a world outside the filter's range keeps every sphere of every ray (list scan)
    const bool mixed_fixed = precision == RT_PRECISION_MIXED_FIXED;
    if (mixed_fixed && !world->filter_ok && sphere_fetch == RT_FETCH_BVH) sphere_fetch = RT_FETCH_SCALAR;
    if ((mixed || mixed_fixed) && sphere_fetch == RT_FETCH_LDS) return fail(RT_ERR_UNSUPPORTED, "RT_PRECISION_MIXED reads its filter records through the scalar cache only");

    const bool f64 = precision == RT_PRECISION_F64 || mixed || mixed_fixed;
    int fetch = sphere_fetch == RT_FETCH_AUTO ? RT_FETCH_SCALAR : sphere_fetch;
    if (fetch == RT_FETCH_LDS && world->n_spheres > (f64 ? rtk::LDS_MAX_SPHERES_F64 : rtk::LDS_MAX_SPHERES_F32))
        return fail(RT_ERR_UNSUPPORTED, "RT_FETCH_LDS holds at most %d spheres at this precision, the world has %d",
                    f64 ? rtk::LDS_MAX_SPHERES_F64 : rtk::LDS_MAX_SPHERES_F32, world->n_spheres);

    const bool queue_mode = precision == RT_PRECISION_F32 || mixed || mixed_fixed;
    int chunks = sample_chunks > 0 && !mixed ? sample_chunks : (queue_mode ? auto_queue_chunks(sample_count, mixed_fixed) : auto_chunks(wk.map.n_tiles, sample_count));
    chunks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(chunks, std::max<int64_t>(1, sample_count))));
    int64_t chunk_samples = sample_count == 0 ? 1 : (sample_count + chunks - 1) / chunks;
    chunks = sample_count == 0 ? 1 : static_cast<int>((sample_count + chunk_samples - 1) / chunk_samples);
    if (static_cast<int64_t>(wk.map.n_tiles) * chunks > (int64_t(1) << 31) - 1) return fail(RT_ERR_UNSUPPORTED, "%d tiles x %d sample chunks exceed 2^31 work items", wk.map.n_tiles, chunks);

    std::lock_guard<std::mutex> lock(world->mu);
    DeviceGuard guard;
    HIP_TRY(guard.enter(world->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t slab_stride = mixed ? static_cast<int64_t>(wk.map.n_tiles) * 64 * 3 : width * height * 3;

    // scratch beyond the queue head: chunk slabs (literal modes) or staged sample colours (mixed mode)
    size_t work_need = 0;
    int64_t window = sample_count; // mixed mode: samples staged per launch
    int halves = 1;                // 2: the windows alternate between two halves of the buffer and overlap
    if (mixed) {
        const size_t per_sample = static_cast<size_t>(slab_stride) * sizeof(double);
        const size_t all = per_sample * static_cast<size_t>(std::max<int64_t>(sample_count, 1));
        const size_t cap = src.fixed ? src.work_bytes : std::max(per_sample, std::min(all, std::min(staging_budget(), world->staging_cap)));
        if (cap < per_sample) return fail(RT_ERR_UNSUPPORTED, "one staged sample of this call needs %zu bytes of scratch, %zu are available", per_sample, cap);
        window = std::max<int64_t>(1, std::min<int64_t>(sample_count, static_cast<int64_t>(cap / per_sample)));
        if (window < sample_count && !src.fixed && cap / 2 >= per_sample) { halves = 2; window = static_cast<int64_t>(cap / 2 / per_sample); }
        if (window < sample_count && window >= unit_samples()) window = window / unit_samples() * unit_samples(); // whole work items
        work_need = per_sample * static_cast<size_t>(window) * static_cast<size_t>(halves);
    } else if (chunks > 1 && !queue_mode) {
        work_need = static_cast<size_t>(chunks - 1) * static_cast<size_t>(slab_stride) * accum_size(precision);
    }
    void *work = nullptr;
    if (work_need > 0) {
        if (src.fixed) {
            if (work_need > src.work_bytes) return fail(RT_ERR_UNSUPPORTED, "the call needs %zu bytes of scratch for its %d sample chunks, the fixed region holds %zu", work_need, chunks, src.work_bytes);
            work = src.work;
        } else {
            if (work_need > world->slab_bytes) {
                if (world->slabs) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(world->slabs)); world->slabs = nullptr; world->slab_bytes = 0; world->slabs_busy = false; }
                hipError_t me = staging_malloc(&world->slabs, work_need);
                // the staged samples are a window that can be any size down to one sample: halve it until it fits
                while (me == hipErrorOutOfMemory && mixed && window > 1) {
                    world->staging_cap = work_need / 2; // later calls start from what fitted
                    window = std::max<int64_t>(1, window / 2);
                    if (window >= unit_samples()) window = window / unit_samples() * unit_samples();
                    work_need = static_cast<size_t>(slab_stride) * sizeof(double) * static_cast<size_t>(window) * static_cast<size_t>(halves);
                    me = staging_malloc(&world->slabs, work_need);
                }
                if (me != hipSuccess) return fail(RT_ERR_DEVICE, "allocating %zu bytes of scratch failed: %s", work_need, hipGetErrorString(me));
                world->slab_bytes = work_need;
            }
            work = world->slabs;
            // the previous user of the buffer may still be running on another stream
            if (world->slabs_busy && world->slabs_stream != st) HIP_TRY(hipStreamWaitEvent(st, world->slabs_done, 0));
        }
    }

    if (fetch == RT_FETCH_BVH && !world->bvh_built) {
        if (src.fixed) return fail(RT_ERR_UNSUPPORTED, "the hierarchy of RT_FETCH_BVH is built and uploaded by the first rt_render_accumulate that asks for it");
        rtk::BvhHost &bvh = world->bvh_host;
        rtk::build_bvh(world->h_rec32.data(), world->h_km.data(), world->h_rad32.data(), world->h_rad32inv.data(), world->h_material.data(), world->n_spheres, bvh);
        // a previous attempt may have failed half way: start from nothing
        for (void **buf : {&world->bvh_nodes, &world->bvh_spheres, &world->bvh_lds_boxes, &world->bvh_lds_links})
            if (*buf) { (void)hipFree(*buf); *buf = nullptr; }
        HIP_TRY(hipMalloc(&world->bvh_nodes, std::max<size_t>(64, bvh.nodes.size() * sizeof(rtk::BvhNode))));
        HIP_TRY(hipMalloc(&world->bvh_spheres, std::max<size_t>(64, bvh.spheres.size() * sizeof(rtk::BvhSphere))));
        if (!bvh.nodes.empty()) HIP_TRY(hipMemcpy(world->bvh_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(rtk::BvhNode), hipMemcpyHostToDevice));
        if (!bvh.spheres.empty()) HIP_TRY(hipMemcpy(world->bvh_spheres, bvh.spheres.data(), bvh.spheres.size() * sizeof(rtk::BvhSphere), hipMemcpyHostToDevice));
        world->bvh_n_nodes = bvh.n_nodes;
        if (!bvh.lds_boxes.empty()) {
            HIP_TRY(hipMalloc(&world->bvh_lds_boxes, bvh.lds_boxes.size() * sizeof(rtk::BvhBoxLds)));
            HIP_TRY(hipMalloc(&world->bvh_lds_links, bvh.lds_links.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(world->bvh_lds_boxes, bvh.lds_boxes.data(), bvh.lds_boxes.size() * sizeof(rtk::BvhBoxLds), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(world->bvh_lds_links, bvh.lds_links.data(), bvh.lds_links.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        std::vector<rtk::BvhBoxLds>().swap(bvh.lds_boxes);
        std::vector<uint32_t>().swap(bvh.lds_links);
        std::vector<rtk::BvhNode>().swap(bvh.nodes);
        std::vector<rtk::BvhSphere>().swap(bvh.spheres);
        world->bvh_built = true;
    }

    // The entry point that allocates nothing may be called while its stream is being captured into a HIP graph: such a
    // call must not synchronise with the host or record the world's timing events (an event recorded during capture
    // cannot be waited for or queried afterwards), and it needs no slot, since its queue head is the fixed one.
    bool capturing = false;
    if (src.fixed) {
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &status) == hipSuccess) capturing = status == hipStreamCaptureStatusActive;
        else (void)hipGetLastError();
    }
    // this call's slot: its own queue head and events; a slot is reused only after its last call has ended
    const int slot_index = static_cast<int>(world->n_calls % CALL_SLOTS);
    CallSlot *slot = capturing ? nullptr : &world->slots[slot_index];
    if (slot != nullptr) {
        if (slot->used) HIP_TRY(hipEventSynchronize(slot->ev_end));
        ++world->n_calls;
    }
    uint32_t *queue = src.queue != nullptr ? src.queue : slot->queue;

    rtk::RenderParams p{};
    p.bvh_nodes = world->bvh_nodes; p.bvh_spheres = world->bvh_spheres; p.bvh_n_nodes = world->bvh_n_nodes;
    if (!bvh_global_only) { p.bvh_lds_boxes = world->bvh_lds_boxes; p.bvh_lds_links = world->bvh_lds_links; }
    if (fetch == RT_FETCH_BVH) {
        const rtk::BvhHost &bvh = world->bvh_host;
        p.bvh_n_big = bvh.n_big;
        for (int a = 0; a < 3; ++a) { p.bvh_origin[a] = bvh.origin[a]; p.bvh_step[a] = bvh.step[a]; }
        static_assert(sizeof(rtk::BvhSphere) == sizeof(p.bvh_big[0]), "big-sphere records are copied bytewise");
        std::memcpy(p.bvh_big, bvh.big, sizeof(rtk::BvhSphere) * static_cast<size_t>(bvh.n_big));
    }
    p.sphere_rec = queue_mode ? world->rec32m : (f64 ? world->rec64 : world->rec32);
    // a world outside the range the filter's error bound needs: the throughput mode keeps every sphere of every ray
    // (the mixed mode has handed such a world to the f64 kernel above)
    p.filter_reach = world->filter_ok ? world->filter_reach : 3.402823466e+38f;
    p.sphere_center = f64 ? world->rec64 : world->rec32;
    p.sphere_radius = f64 ? world->rad64 : (precision == RT_PRECISION_F32 ? world->rad32inv : world->rad32);
    p.sphere_material = world->sphere_material;
    p.material_rec = f64 ? world->mat64 : world->mat32;
    p.material_type = world->material_type;
    p.n_spheres = world->n_padded;
    p.width = static_cast<int32_t>(width); p.height = static_cast<int32_t>(height);
    p.min_x = wk.min_x; p.min_y = wk.min_y; p.max_x = wk.max_x; p.max_y = wk.max_y;
    p.map = wk.map;
    p.sample_begin = static_cast<uint32_t>(sample_begin);
    p.sample_count = static_cast<uint32_t>(sample_count);
    p.chunk_samples = static_cast<uint32_t>(chunk_samples);
    p.n_items = static_cast<uint32_t>(static_cast<int64_t>(wk.map.n_tiles) * chunks);
    p.queue = queue;
    p.max_bounces = static_cast<int32_t>(cam->max_ray_bounces);
    seed_keys(seed, p.k0, p.k1);
    p.accum = d_accum;
    p.slabs = work;
    p.slab_stride = slab_stride;
    p.stats = reinterpret_cast<unsigned long long *>(d_stats);
    pack_camera(cam, p.cam32, p.cam64);

    rtk::ReduceParams r{};
    r.accum = d_accum; r.slabs = work; r.slab_stride = slab_stride; r.chunks = chunks;
    r.words_per_pixel = mixed_fixed ? 6 : 3;
    r.width = p.width; r.min_x = p.min_x; r.min_y = p.min_y; r.max_x = p.max_x; r.max_y = p.max_y; r.map = p.map;
    if (slot != nullptr) {
        HIP_TRY(hipEventRecord(slot->ev_begin, st));
        slot->used = true; // from here on something of this call may be in flight
        world->last_slot = slot_index;
    }
    if (mixed && halves == 2) {
        // Overlapping windows.  Window k traces into half k % 2 of the buffer on stream k % 2 and is followed there by its
        // ordered sum, which first waits for the sum of window k - 1 (the accumulators are added to in sample order).
        // Window k + 1 starts on the other stream as soon as that stream's previous sum is done, i.e. while window k is
        // still running: its workgroups take the compute units window k's last waves free.  A half is written again
        // (window k + 2) only after its sum (window k), in stream order.  The caller's stream forks into the two and
        // joins them again, so everything the call does lies between the events recorded on it.
        for (int k = 0; k < 2; ++k) {
            if (world->win_stream[k] == nullptr) HIP_TRY(hipStreamCreateWithFlags(&world->win_stream[k], hipStreamNonBlocking));
            if (world->win_sum[k] == nullptr) HIP_TRY(hipEventCreateWithFlags(&world->win_sum[k], hipEventDisableTiming));
            if (world->win_join[k] == nullptr) HIP_TRY(hipEventCreateWithFlags(&world->win_join[k], hipEventDisableTiming));
        }
        if (world->win_fork == nullptr) HIP_TRY(hipEventCreateWithFlags(&world->win_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(world->win_fork, st));
        for (int k = 0; k < 2; ++k) HIP_TRY(hipStreamWaitEvent(world->win_stream[k], world->win_fork, 0));
        const size_t half_elems = static_cast<size_t>(slab_stride) * static_cast<size_t>(window);
        int k = 0;
        for (int64_t w0 = 0; w0 < sample_count; w0 += window, ++k) {
            const int side = k & 1;
            hipStream_t ws = world->win_stream[side];
            const int64_t wc = std::min<int64_t>(window, sample_count - w0);
            const int64_t unit = unit_samples();
            const int wchunks = static_cast<int>(std::max<int64_t>(1, (wc + unit - 1) / unit));
            p.sample_begin = static_cast<uint32_t>(sample_begin + w0);
            p.sample_count = static_cast<uint32_t>(wc);
            p.chunk_samples = static_cast<uint32_t>(unit);
            p.n_items = static_cast<uint32_t>(static_cast<int64_t>(wk.map.n_tiles) * wchunks);
            p.queue = queue + 8 * side;
            p.slabs = static_cast<double *>(work) + half_elems * static_cast<size_t>(side);
            r.slabs = p.slabs;
            r.chunks = static_cast<int32_t>(wc);
            HIP_TRY(rtk::launch_clear_queue(p.queue, ws));
            HIP_TRY(rtk::launch_render(p, rtk::ARITH_MIXED, fetch, wchunks, world->resident_blocks, ws));
            if (k > 0) HIP_TRY(hipStreamWaitEvent(ws, world->win_sum[side ^ 1], 0));
            HIP_TRY(rtk::launch_order_sum(r, k == 0, ws));
            HIP_TRY(hipEventRecord(world->win_sum[side], ws));
        }
        for (int j = 0; j < 2; ++j) {
            HIP_TRY(hipEventRecord(world->win_join[j], world->win_stream[j]));
            HIP_TRY(hipStreamWaitEvent(st, world->win_join[j], 0));
        }
    } else if (mixed) {
        // samples in windows of what the staging buffer holds: trace a window (any order), then add its
        // colours to the accumulators in sample order
        for (int64_t w0 = 0; w0 < std::max<int64_t>(sample_count, 1); w0 += window) {
            const int64_t wc = std::min<int64_t>(window, sample_count - w0);
            const int64_t unit = unit_samples();
            const int wchunks = static_cast<int>(std::max<int64_t>(1, (wc + unit - 1) / unit));
            p.sample_begin = static_cast<uint32_t>(sample_begin + w0);
            p.sample_count = static_cast<uint32_t>(wc);
            p.chunk_samples = static_cast<uint32_t>(unit);
            p.n_items = static_cast<uint32_t>(static_cast<int64_t>(wk.map.n_tiles) * wchunks);
            r.chunks = static_cast<int32_t>(wc);
            HIP_TRY(rtk::launch_clear_queue(queue, st));
            if (wc > 0) HIP_TRY(rtk::launch_render(p, rtk::ARITH_MIXED, fetch, wchunks, world->resident_blocks, st));
            HIP_TRY(rtk::launch_order_sum(r, w0 == 0, st));
        }
    } else {
        if (queue_mode) HIP_TRY(rtk::launch_clear(r, queue, st));
        HIP_TRY(rtk::launch_render(p, precision, fetch, chunks, world->resident_blocks, st));
        if (chunks > 1 && !queue_mode) HIP_TRY(rtk::launch_reduce(r, precision, st));
    }
    if (slot != nullptr) HIP_TRY(hipEventRecord(slot->ev_end, st));
    if (work != nullptr && !src.fixed) {
        HIP_TRY(hipEventRecord(world->slabs_done, st));
        world->slabs_busy = true;
        world->slabs_stream = st;
    }
    return RT_OK;
}

} // namespace

extern "C" {

int32_t rt_render_accumulate(rt_world *world, const rt_camera_render_data *cam, int64_t width, int64_t height, rt_rect region,
                             int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                             int64_t sample_begin, int64_t sample_count, uint64_t seed, int32_t precision, int32_t sphere_fetch,
                             int32_t sample_chunks, void *d_accum, uint64_t *d_stats, void *stream) {
    return accumulate_impl(world, cam, width, height, region, strip_rows, strip_index, strip_count, sample_begin, sample_count, seed, precision,
                           sphere_fetch, sample_chunks, d_accum, d_stats, stream, ScratchSource{});
}

int32_t rt_render_accumulate_static(rt_world *world, const rt_camera_render_data *cam, int64_t width, int64_t height, rt_rect region,
                                    int64_t strip_rows, int32_t strip_index, int32_t strip_count,
                                    int64_t sample_begin, int64_t sample_count, uint64_t seed, int32_t precision, int32_t sphere_fetch,
                                    int32_t sample_chunks, void *d_accum, uint64_t *d_stats, void *stream) {
    if (world == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "world is NULL");
    // the per-device address of the static scratch array was looked up when the world was created
    if (world->static_scratch == nullptr) return fail(RT_ERR_DEVICE, "the static scratch region of device %d is not available", world->device);
    ScratchSource src;
    src.queue = static_cast<uint32_t *>(world->static_scratch);
    src.work = static_cast<unsigned char *>(world->static_scratch) + rtk::STATIC_SCRATCH_HEAD_BYTES;
    src.work_bytes = rtk::STATIC_SCRATCH_BYTES - rtk::STATIC_SCRATCH_HEAD_BYTES;
    src.fixed = true;
    return accumulate_impl(world, cam, width, height, region, strip_rows, strip_index, strip_count, sample_begin, sample_count, seed, precision,
                           sphere_fetch, sample_chunks, d_accum, d_stats, stream, src);
}

int64_t rt_static_scratch_bytes(void) { return static_cast<int64_t>(rtk::STATIC_SCRATCH_BYTES - rtk::STATIC_SCRATCH_HEAD_BYTES); }

int64_t rt_world_last_kernel_ns(rt_world *world) {
    if (world == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "world is NULL");
    std::lock_guard<std::mutex> lock(world->mu);
    if (world->last_slot < 0) return fail(RT_ERR_INVALID_ARGUMENT, "no render has been launched on this world");
    DeviceGuard guard;
    if (guard.enter(world->device) != hipSuccess) return fail(RT_ERR_DEVICE, "hipSetDevice failed");
    const CallSlot &slot = world->slots[world->last_slot];
    hipError_t e = hipEventSynchronize(slot.ev_end);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "hipEventSynchronize failed: %s", hipGetErrorString(e));
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, slot.ev_begin, slot.ev_end);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "hipEventElapsedTime failed: %s", hipGetErrorString(e));
    return static_cast<int64_t>(static_cast<double>(ms) * 1e6);
}

int32_t rt_finalize(rt_world *world, const void *d_accum, int32_t precision, int64_t width, int64_t height, rt_rect region,
                    int64_t strip_rows, int32_t strip_index, int32_t strip_count, int64_t total_samples,
                    rt_v3 *d_image, rt_v3 *d_linear, void *stream) {
    if (world == nullptr || d_accum == nullptr || d_image == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "world, d_accum or d_image is NULL");
    if (!precision_ok(precision)) return fail(RT_ERR_INVALID_ARGUMENT, "unknown precision %d", precision);
    if (total_samples <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "total_samples %lld <= 0", (long long)total_samples);
    Work wk;
    int32_t rc = plan_work(width, height, region, strip_rows, strip_index, strip_count, wk);
    if (rc != RT_OK) return rc;
    if (wk.map.n_tiles == 0) return RT_OK;
    DeviceGuard guard;
    HIP_TRY(guard.enter(world->device));
    rtk::FinalizeParams f{};
    f.accum = d_accum;
    f.image = reinterpret_cast<double *>(d_image);
    f.linear = reinterpret_cast<double *>(d_linear);
    f.total_samples = total_samples;
    f.width = static_cast<int32_t>(width);
    f.min_x = wk.min_x; f.min_y = wk.min_y; f.max_x = wk.max_x; f.max_y = wk.max_y; f.map = wk.map;
    HIP_TRY(rtk::launch_finalize(f, precision, static_cast<hipStream_t>(stream)));
    return RT_OK;
}

int32_t rt_finalize_ppm(rt_world *world, const void *d_accum, int32_t precision, int64_t width, int64_t height,
                        int64_t strip_rows, int32_t strip_index, int32_t strip_count, int64_t total_samples,
                        char *d_text, void *stream) {
    if (world == nullptr || d_accum == nullptr || d_text == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "world, d_accum or d_text is NULL");
    if (!precision_ok(precision)) return fail(RT_ERR_INVALID_ARGUMENT, "unknown precision %d", precision);
    if (total_samples <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "total_samples %lld <= 0", (long long)total_samples);
    if ((reinterpret_cast<uintptr_t>(d_text) & 3u) != 0) return fail(RT_ERR_INVALID_ARGUMENT, "d_text must be 4-byte aligned");
    Work wk;
    const rt_rect whole = {0, 0, width, height};
    int32_t rc = plan_work(width, height, whole, strip_rows, strip_index, strip_count, wk);
    if (rc != RT_OK) return rc;
    if (wk.map.n_tiles == 0) return RT_OK;
    DeviceGuard guard;
    HIP_TRY(guard.enter(world->device));
    rtk::PpmParams f{};
    f.accum = d_accum;
    f.text = reinterpret_cast<uint32_t *>(d_text);
    f.total_samples = total_samples;
    f.width = static_cast<int32_t>(width);
    f.min_x = wk.min_x; f.min_y = wk.min_y; f.max_x = wk.max_x; f.max_y = wk.max_y; f.map = wk.map;
    HIP_TRY(rtk::launch_finalize_ppm(f, precision, static_cast<hipStream_t>(stream)));
    return RT_OK;
}

} // extern "C"

namespace {

// What a one-shot call needs on a device, kept between calls: the reference's benchmark calls
// render_region 100 times on one world (main_test.odin:25-32), and creating the world, the buffers
// and the stream anew each time would cost more than the kernels.  One entry per device ordinal,
// used under that device's mutex; the world is reused when the caller's spheres and materials are
// byte for byte those it was made from.  rt_release_cache frees everything.
struct DeviceCache {
    rt_world *world = nullptr;
    std::vector<unsigned char> scene;   // the spheres' bytes followed by the materials'
    int64_t n_spheres = -1, n_materials = -1;
    void *accum = nullptr; size_t accum_bytes = 0;
    void *image = nullptr; size_t image_bytes = 0;
    void *linear = nullptr; size_t linear_bytes = 0;
    void *text = nullptr; size_t text_bytes = 0;
    uint64_t *stats = nullptr;
    hipStream_t stream = nullptr;
    void *pinned = nullptr; size_t pinned_bytes = 0; // page-locked host staging for the copy back
};
DeviceCache g_cache[MAX_DEVICES];
constexpr size_t PINNED_STAGING_MAX = static_cast<size_t>(1) << 30; // larger results are copied straight into the caller's memory
constexpr size_t ONE_SHOT_STAGING_KEPT = static_cast<size_t>(256) << 20; // a cached world's staging buffer above this is freed when the call returns

hipError_t ensure_bytes(void **buf, size_t *have, size_t need) {
    if (*buf != nullptr && *have >= need) return hipSuccess;
    if (*buf != nullptr) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    hipError_t e = hipMalloc(buf, std::max<size_t>(need, 64));
    if (e == hipSuccess) *have = need;
    return e;
}

void release_cache_entry(DeviceCache &c) {
    if (c.stream) { (void)hipStreamSynchronize(c.stream); (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    rt_world_destroy(c.world); c.world = nullptr;
    c.scene.clear(); c.n_spheres = c.n_materials = -1;
    for (void **b : {&c.accum, &c.image, &c.linear, &c.text, reinterpret_cast<void **>(&c.stats)}) if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (c.pinned) { (void)hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_bytes = 0; }
    c.accum_bytes = c.image_bytes = c.linear_bytes = c.text_bytes = 0;
}

// rt_render_region and rt_render_ppm: host buffers in, host buffers out.  Exactly one of
// image_data (f64 image of `region`) and ppm_body (12 bytes of text per pixel of the whole
// image) is non-NULL.
int32_t render_to_host(rt_v3 *image_data, char *ppm_body, int64_t width, int64_t height, rt_rect region, const rt_camera_render_data *cam,
                       const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials,
                       const rt_options *opts) {
    rt_options o;
    if (opts) o = *opts; else rt_options_default(&o);
    if ((image_data == nullptr && ppm_body == nullptr) || cam == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "output buffer or cam is NULL");
    if (!precision_ok(o.precision)) return fail(RT_ERR_INVALID_ARGUMENT, "unknown precision %d", o.precision);
    if (o.rng_mode != RT_RNG_STOCHASTIC && o.rng_mode != RT_RNG_DETERMINISTIC) return fail(RT_ERR_INVALID_ARGUMENT, "unknown rng_mode %d", o.rng_mode);
    if ((o.flags & ~RT_FLAG_PRINT_PROGRESS) != 0) return fail(RT_ERR_INVALID_ARGUMENT, "unknown bits in flags 0x%x", static_cast<unsigned>(o.flags));
    // the reference divides by samples_per_pixel (main.odin:512); zero or negative counts have no meaningful image
    if (cam->samples_per_pixel < 1) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel %lld < 1", (long long)cam->samples_per_pixel);
    int32_t rc = validate_scene(spheres, n_spheres, materials, n_materials);
    if (rc != RT_OK) return rc;
    Work probe;
    rc = plan_work(width, height, region, rtk::TILE, 0, 1, probe);
    if (rc != RT_OK) return rc;
    if (o.stats) std::memset(o.stats, 0, sizeof(uint64_t) * RT_STATS_COUNT);
    if (probe.map.n_tiles == 0) return RT_OK; // empty region: nothing to render, nothing written

    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    int first = o.device;
    if (first < 0) HIP_TRY(hipGetDevice(&first));
    const int count = std::max(1, o.device_count);
    // RT_MI355X_WRAP_DEVICES=1 maps device ordinals modulo the visible count, so that the strip logic of
    // device_count > 1 can be exercised on a machine with fewer GPUs (every "device" then has its own stream,
    // world and buffers on a shared GPU)
    const char *wrap_env = std::getenv("RT_MI355X_WRAP_DEVICES");
    const bool wrap = wrap_env != nullptr && wrap_env[0] == '1' && n_dev > 0;
    if (!wrap && (first >= n_dev || first + count > n_dev)) return fail(RT_ERR_DEVICE, "devices [%d, %d) requested, %d visible", first, first + count, n_dev);
    if (n_dev > MAX_DEVICES || count > MAX_DEVICES) return fail(RT_ERR_UNSUPPORTED, "more than %d devices", MAX_DEVICES);
    auto ordinal = [&](int g) { return wrap ? (first + g) % n_dev : first + g; };

    uint64_t seed = o.seed;
    if (o.rng_mode == RT_RNG_STOCHASTIC) {
        const uint64_t tick = static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count());
        seed ^= (tick * 0x9e3779b97f4a7c15ull) ^ ((g_call_counter.fetch_add(1) + 1) * 0xbf58476d1ce4e5b9ull);
    }

    struct PerDevice {
        int device = 0; bool cached = false;
        rt_world *world = nullptr; void *accum = nullptr; double *image = nullptr; double *linear = nullptr; char *text = nullptr;
        uint64_t *stats = nullptr; hipStream_t stream = nullptr;
        void *pinned = nullptr; size_t pinned_bytes = 0; // host staging of the cached device (NULL: copy straight to the caller's memory)
    };
    std::vector<PerDevice> devs(count);
    const size_t n_px = static_cast<size_t>(width) * static_cast<size_t>(height);
    const int64_t strip_rows = rtk::TILE; // 8-row strips dealt round-robin over the devices
    int32_t result = RT_OK;
    auto cleanup = [&]() {
        for (auto &d : devs) {
            (void)hipSetDevice(d.device);
            if (d.stream) (void)hipStreamSynchronize(d.stream);
            if (d.cached) {
                // kept for the next call -- except a large staging buffer of the ordered mixed mode: a host that calls once
                // per frame should not hold gigabytes between frames (allocating it again costs a few ms of a frame that
                // takes a hundred; the small buffers of a benchmark loop over a small image stay)
                rt_world *w = d.world;
                if (w != nullptr && w->slabs != nullptr && w->slab_bytes > ONE_SHOT_STAGING_KEPT) {
                    std::lock_guard<std::mutex> wl(w->mu);
                    (void)hipFree(w->slabs);
                    w->slabs = nullptr; w->slab_bytes = 0; w->slabs_busy = false;
                }
                continue;
            }
            if (d.stream) (void)hipStreamDestroy(d.stream);
            if (d.accum) (void)hipFree(d.accum);
            if (d.image) (void)hipFree(d.image);
            if (d.text) (void)hipFree(d.text);
            if (d.linear) (void)hipFree(d.linear);
            if (d.stats) (void)hipFree(d.stats);
            rt_world_destroy(d.world);
        }
    };
    int prev_device = 0;
    HIP_TRY(hipGetDevice(&prev_device));
    std::vector<std::unique_lock<std::mutex>> locks;
    {
        std::vector<int> used;
        for (int g = 0; g < count; ++g) used.push_back(ordinal(g));
        std::sort(used.begin(), used.end());
        used.erase(std::unique(used.begin(), used.end()), used.end());
        for (int dev : used) locks.emplace_back(device_mutex(dev)); // ascending order: no lock-order inversion between calls
    }
    const size_t scene_bytes = (static_cast<size_t>(n_spheres) * sizeof(rt_sphere)) + static_cast<size_t>(n_materials) * sizeof(rt_material);
    auto same_scene = [&](const DeviceCache &c) {
        if (c.world == nullptr || c.n_spheres != n_spheres || c.n_materials != n_materials || c.scene.size() != scene_bytes) return false;
        const size_t sb = static_cast<size_t>(n_spheres) * sizeof(rt_sphere);
        return (sb == 0 || std::memcmp(c.scene.data(), spheres, sb) == 0) &&
               (scene_bytes == sb || std::memcmp(c.scene.data() + sb, materials, scene_bytes - sb) == 0);
    };

    // launch everything on every device, then collect
    bool cache_taken[MAX_DEVICES] = {};
#ifdef RT_DIAG_NO_CACHE
    for (bool &t : cache_taken) t = true; // diagnostic build: every call sets up and tears down, to measure what the cache saves
#endif
    for (int g = 0; g < count && result == RT_OK; ++g) {
        PerDevice &d = devs[g];
        d.device = ordinal(g);
        hipError_t e = hipSetDevice(d.device);
        const size_t accum_need = n_px * 3 * accum_size(o.precision), image_need = n_px * 3 * sizeof(double), text_need = n_px * 12;
        if (e == hipSuccess && !cache_taken[d.device]) {
            // the first strip set on this device uses (and fills) the device's cache
            cache_taken[d.device] = true;
            d.cached = true;
            DeviceCache &c = g_cache[d.device];
            if (c.stream == nullptr) e = hipStreamCreate(&c.stream);
            if (e == hipSuccess) e = ensure_bytes(&c.accum, &c.accum_bytes, accum_need);
            if (e == hipSuccess && image_data) e = ensure_bytes(&c.image, &c.image_bytes, image_need);
            if (e == hipSuccess && ppm_body) e = ensure_bytes(&c.text, &c.text_bytes, text_need);
            if (e == hipSuccess && o.linear_out && image_data) e = ensure_bytes(&c.linear, &c.linear_bytes, image_need);
            if (e == hipSuccess && c.stats == nullptr) e = hipMalloc(reinterpret_cast<void **>(&c.stats), sizeof(uint64_t) * RT_STATS_COUNT);
            {
                // staging for the rows of the region (whole image rows), twice with linear_out, plus the counters
                const size_t rows = static_cast<size_t>(region.one_past_max_y - region.min_y);
                const size_t stage_need = rows * static_cast<size_t>(width) * (ppm_body ? 12 : sizeof(rt_v3)) * ((o.linear_out && image_data) ? 2 : 1) + sizeof(uint64_t) * RT_STATS_COUNT;
                if (e == hipSuccess && stage_need <= PINNED_STAGING_MAX && stage_need > c.pinned_bytes) {
                    if (c.pinned) { (void)hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_bytes = 0; }
                    if (hipHostMalloc(&c.pinned, stage_need, hipHostMallocDefault) == hipSuccess) c.pinned_bytes = stage_need;
                    else { c.pinned = nullptr; (void)hipGetLastError(); } // no page-locked memory: the direct copy still works
                }
                if (stage_need <= c.pinned_bytes) { d.pinned = c.pinned; d.pinned_bytes = c.pinned_bytes; }
            }
            d.stream = c.stream; d.accum = c.accum; d.image = static_cast<double *>(c.image); d.text = static_cast<char *>(c.text);
            d.linear = (o.linear_out && image_data) ? static_cast<double *>(c.linear) : nullptr; d.stats = c.stats;
            if (e == hipSuccess && !same_scene(c)) {
                rt_world_destroy(c.world); c.world = nullptr; c.n_spheres = c.n_materials = -1;
                result = rt_world_create(spheres, n_spheres, materials, n_materials, d.device, &c.world);
                if (result == RT_OK) {
                    c.scene.resize(scene_bytes);
                    const size_t sb = static_cast<size_t>(n_spheres) * sizeof(rt_sphere);
                    if (sb) std::memcpy(c.scene.data(), spheres, sb);
                    if (scene_bytes > sb) std::memcpy(c.scene.data() + sb, materials, scene_bytes - sb);
                    c.n_spheres = n_spheres; c.n_materials = n_materials;
                }
            }
            d.world = c.world;
        } else if (e == hipSuccess) {
            e = hipStreamCreate(&d.stream);
            if (e == hipSuccess) e = hipMalloc(&d.accum, std::max<size_t>(accum_need, 64));
            if (e == hipSuccess && image_data) e = hipMalloc(reinterpret_cast<void **>(&d.image), image_need);
            if (e == hipSuccess && ppm_body) e = hipMalloc(reinterpret_cast<void **>(&d.text), text_need);
            if (e == hipSuccess && o.linear_out && image_data) e = hipMalloc(reinterpret_cast<void **>(&d.linear), image_need);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d.stats), sizeof(uint64_t) * RT_STATS_COUNT);
            if (e == hipSuccess) result = rt_world_create(spheres, n_spheres, materials, n_materials, d.device, &d.world);
        }
        if (e == hipSuccess && result == RT_OK) e = hipMemsetAsync(d.stats, 0, sizeof(uint64_t) * RT_STATS_COUNT, d.stream);
        if (e != hipSuccess) { result = fail(RT_ERR_DEVICE, "device %d set-up failed: %s", d.device, hipGetErrorString(e)); break; }
        if (result != RT_OK) break;
        result = rt_render_accumulate(d.world, cam, width, height, region, strip_rows, g, count, 0, cam->samples_per_pixel, seed, o.precision,
                                      o.sphere_fetch, o.sample_chunks, d.accum, d.stats, d.stream);
        if (result != RT_OK) break;
        if (image_data)
            result = rt_finalize(d.world, d.accum, o.precision, width, height, region, strip_rows, g, count, cam->samples_per_pixel,
                                 reinterpret_cast<rt_v3 *>(d.image), reinterpret_cast<rt_v3 *>(d.linear), d.stream);
        else
            result = rt_finalize_ppm(d.world, d.accum, o.precision, width, height, strip_rows, g, count, cam->samples_per_pixel, d.text, d.stream);
    }
    // Copy each device's strips of the region back: rows [y0, y1), columns of the region.  A device with a pinned
    // staging buffer gets its rows (whole image rows, which are contiguous on the device) by asynchronous copies on
    // its own stream, all devices at once, and the host then copies the region's columns out of the staging buffer;
    // copying into the caller's pageable memory directly costs several times as much for small images.
    const size_t pitch = static_cast<size_t>(width) * sizeof(rt_v3);
    const size_t row_bytes = static_cast<size_t>(region.one_past_max_x - region.min_x) * sizeof(rt_v3);
    auto for_each_span = [&](int g, auto &&fn) { // the row ranges device g owns, in image order
        for (int64_t y0 = region.min_y + g * strip_rows; y0 < region.one_past_max_y; y0 += strip_rows * count) {
            const int64_t y1 = std::min<int64_t>(region.one_past_max_y, count == 1 ? region.one_past_max_y : y0 + strip_rows);
            if (!fn(y0, y1)) return false;
            if (count == 1) break;
        }
        return true;
    };
    const size_t px_bytes = ppm_body ? 12 : sizeof(rt_v3);
    for (int g = 0; g < count && result == RT_OK; ++g) { // enqueue
        PerDevice &d = devs[g];
        if (d.pinned == nullptr) continue;
        hipError_t e = hipSetDevice(d.device);
        size_t at = 0; // bytes staged so far
        const char *src = ppm_body ? d.text : reinterpret_cast<const char *>(d.image);
        for (int pass = 0; pass < ((o.linear_out && image_data) ? 2 : 1) && e == hipSuccess; ++pass) {
            if (pass == 1) src = reinterpret_cast<const char *>(d.linear);
            for_each_span(g, [&](int64_t y0, int64_t y1) {
                const size_t bytes = static_cast<size_t>(y1 - y0) * static_cast<size_t>(width) * px_bytes;
                e = hipMemcpyAsync(static_cast<char *>(d.pinned) + at, src + static_cast<size_t>(y0) * static_cast<size_t>(width) * px_bytes, bytes, hipMemcpyDeviceToHost, d.stream);
                at += bytes;
                return e == hipSuccess;
            });
        }
        if (e == hipSuccess && o.stats) e = hipMemcpyAsync(static_cast<char *>(d.pinned) + d.pinned_bytes - sizeof(uint64_t) * RT_STATS_COUNT, d.stats, sizeof(uint64_t) * RT_STATS_COUNT, hipMemcpyDeviceToHost, d.stream);
        if (e != hipSuccess) result = fail(RT_ERR_DEVICE, "collecting from device %d failed: %s", d.device, hipGetErrorString(e));
    }
    for (int g = 0; g < count && result == RT_OK; ++g) { // collect
        PerDevice &d = devs[g];
        hipError_t e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
        uint64_t h[RT_STATS_COUNT] = {0};
        if (e == hipSuccess && d.pinned != nullptr) {
            size_t at = 0;
            for (int pass = 0; pass < ((o.linear_out && image_data) ? 2 : 1); ++pass) {
                char *dst = ppm_body ? ppm_body : reinterpret_cast<char *>(pass == 0 ? image_data : o.linear_out);
                for_each_span(g, [&](int64_t y0, int64_t y1) {
                    const char *from = static_cast<const char *>(d.pinned) + at;
                    if (ppm_body || row_bytes == pitch) { // whole rows
                        std::memcpy(dst + static_cast<size_t>(y0) * static_cast<size_t>(width) * px_bytes, from, static_cast<size_t>(y1 - y0) * static_cast<size_t>(width) * px_bytes);
                    } else {
                        for (int64_t y = y0; y < y1; ++y)
                            std::memcpy(dst + (static_cast<size_t>(y) * static_cast<size_t>(width) + static_cast<size_t>(region.min_x)) * sizeof(rt_v3),
                                        from + (static_cast<size_t>(y - y0) * static_cast<size_t>(width) + static_cast<size_t>(region.min_x)) * sizeof(rt_v3), row_bytes);
                    }
                    at += static_cast<size_t>(y1 - y0) * static_cast<size_t>(width) * px_bytes;
                    return true;
                });
            }
            if (o.stats) std::memcpy(h, static_cast<const char *>(d.pinned) + d.pinned_bytes - sizeof h, sizeof h);
        } else if (e == hipSuccess) {
            for_each_span(g, [&](int64_t y0, int64_t y1) {
                const size_t off = static_cast<size_t>(y0) * static_cast<size_t>(width) + static_cast<size_t>(region.min_x);
                if (ppm_body) { // whole rows of text are contiguous
                    const size_t row0 = static_cast<size_t>(y0) * static_cast<size_t>(width) * 12;
                    e = hipMemcpy(ppm_body + row0, d.text + row0, static_cast<size_t>(y1 - y0) * static_cast<size_t>(width) * 12, hipMemcpyDeviceToHost);
                    return e == hipSuccess;
                }
                e = hipMemcpy2D(image_data + off, pitch, reinterpret_cast<rt_v3 *>(d.image) + off, pitch, row_bytes, static_cast<size_t>(y1 - y0), hipMemcpyDeviceToHost);
                if (e == hipSuccess && o.linear_out)
                    e = hipMemcpy2D(o.linear_out + off, pitch, reinterpret_cast<rt_v3 *>(d.linear) + off, pitch, row_bytes, static_cast<size_t>(y1 - y0), hipMemcpyDeviceToHost);
                return e == hipSuccess;
            });
            if (e == hipSuccess && o.stats) e = hipMemcpy(h, d.stats, sizeof h, hipMemcpyDeviceToHost);
        }
        if (e == hipSuccess && o.stats) {
            o.stats[RT_STAT_SEGMENTS] += h[RT_STAT_SEGMENTS];
            o.stats[RT_STAT_SAMPLES] += h[RT_STAT_SAMPLES];
            o.stats[RT_STAT_WAVE_ITERATIONS] += h[RT_STAT_WAVE_ITERATIONS];
            o.stats[RT_STAT_BVH_NODES] += h[RT_STAT_BVH_NODES];
            o.stats[RT_STAT_BVH_TESTS] += h[RT_STAT_BVH_TESTS];
            o.stats[7] += h[7]; // diagnostic builds only
            const int64_t ns = rt_world_last_kernel_ns(d.world);
            if (ns > 0) o.stats[RT_STAT_KERNEL_NS] = std::max<uint64_t>(o.stats[RT_STAT_KERNEL_NS], static_cast<uint64_t>(ns));
        }
        if (e != hipSuccess) result = fail(RT_ERR_DEVICE, "collecting from device %d failed: %s", d.device, hipGetErrorString(e));
    }
    if (result == RT_OK && o.stats)
        o.stats[RT_STAT_SPHERE_TESTS] = (o.sphere_fetch == RT_FETCH_BVH || o.sphere_fetch == RT_FETCH_BVH_GLOBAL) ? o.stats[RT_STAT_BVH_TESTS] : o.stats[RT_STAT_SEGMENTS] * static_cast<uint64_t>(n_spheres);
    // $print_progress of render_region: the reference's closing line (main.odin:590)
    if (result == RT_OK && (o.flags & RT_FLAG_PRINT_PROGRESS) != 0) std::fputs("\rDone.                   \n", stderr);
    const std::string keep = g_error;
    cleanup();
    (void)hipSetDevice(prev_device);
    if (result != RT_OK) g_error = keep;
    return result;
}

} // namespace

extern "C" {

int32_t rt_render_region(rt_v3 *image_data, int64_t width, int64_t height, rt_rect region, const rt_camera_render_data *cam,
                         const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials,
                         const rt_options *opts) {
    if (image_data == nullptr) return fail(RT_ERR_INVALID_ARGUMENT, "image_data is NULL");
    return render_to_host(image_data, nullptr, width, height, region, cam, spheres, n_spheres, materials, n_materials, opts);
}

int64_t rt_render_ppm(int64_t width, int64_t height, const rt_camera_render_data *cam,
                      const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials,
                      const rt_options *opts, char *dst, int64_t capacity) {
    if (width < 0 || height < 0) return fail(RT_ERR_INVALID_ARGUMENT, "negative image size");
    // the limit of plan_work, applied before any size is computed from the arguments
    if (width > (int64_t(1) << 30) || height > (int64_t(1) << 30) || width * height > (int64_t(1) << 30)) return fail(RT_ERR_UNSUPPORTED, "image of %lld x %lld pixels exceeds 2^30", (long long)width, (long long)height);
    char header[64];
    const int header_len = std::snprintf(header, sizeof header, "P3\n%lld %lld\n255\n", (long long)width, (long long)height);
    const int64_t need = header_len + width * height * 12;
    if (dst == nullptr || capacity < need) return need;
    if (opts && opts->linear_out) return fail(RT_ERR_INVALID_ARGUMENT, "linear_out is not available from rt_render_ppm");
    std::memcpy(dst, header, static_cast<size_t>(header_len));
    if (width * height == 0) return need;
    const rt_rect whole = {0, 0, width, height};
    const int32_t rc = render_to_host(nullptr, dst + header_len, width, height, whole, cam, spheres, n_spheres, materials, n_materials, opts);
    return rc == RT_OK ? need : rc;
}

int32_t rt_release_cache(void) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) return RT_OK; // no device: nothing was ever cached
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int dev = 0; dev < std::min(n_dev, MAX_DEVICES); ++dev) {
        std::lock_guard<std::mutex> lock(device_mutex(dev));
        DeviceCache &c = g_cache[dev];
        if (c.world == nullptr && c.stream == nullptr && c.accum == nullptr) continue;
        (void)hipSetDevice(dev);
        release_cache_entry(c);
    }
    (void)hipSetDevice(prev);
    return RT_OK;
}

int32_t rt_bvh_validate(const rt_sphere *spheres, int64_t n_spheres, int64_t *n_nodes, int64_t *max_leaf) {
    if (n_spheres < 0 || (n_spheres > 0 && spheres == nullptr)) return fail(RT_ERR_INVALID_ARGUMENT, "spheres is NULL or the count negative");
    if (n_spheres > (1 << 24)) return fail(RT_ERR_UNSUPPORTED, "more than 2^24 spheres");
    const size_t ns = static_cast<size_t>(n_spheres);
    std::vector<float> rec(4 * ns), k(ns), rad(ns), inv(ns);
    std::vector<int32_t> mat(ns, 0);
    for (size_t i = 0; i < ns; ++i) {
        const float r = static_cast<float>(spheres[i].radius);
        rec[4 * i] = static_cast<float>(spheres[i].center.x); rec[4 * i + 1] = static_cast<float>(spheres[i].center.y);
        rec[4 * i + 2] = static_cast<float>(spheres[i].center.z); rec[4 * i + 3] = r * r;
        rad[i] = r; inv[i] = 1.0f / r; k[i] = 0.0f;
    }
    rtk::BvhHost bvh;
    rtk::build_bvh(rec.data(), k.data(), rad.data(), inv.data(), mat.data(), static_cast<int32_t>(n_spheres), bvh);
    int64_t worst = 0;
    std::vector<int> seen(ns, 0);
    auto corner = [&](uint16_t q, int a) { return bvh.origin[a] + static_cast<float>(q) * bvh.step[a]; };
    for (int oct = 0; oct < 8; ++oct) {
        const rtk::BvhNode *nodes = bvh.nodes.data() + static_cast<size_t>(oct) * bvh.n_nodes;
        std::fill(seen.begin(), seen.end(), 0);
        for (int g = 0; g < bvh.n_big; ++g) {
            if (bvh.big[g].index < 0 || bvh.big[g].index >= static_cast<int>(ns) || seen[bvh.big[g].index]++) return fail(RT_ERR_DEVICE, "big sphere %d appears twice or is out of range", bvh.big[g].index);
            if (g > 0 && bvh.big[g - 1].index >= bvh.big[g].index) return fail(RT_ERR_DEVICE, "big spheres are not in list order");
        }
        // every node: skip link ahead and inside the array; boxes of a subtree inside the root of that subtree
        for (int32_t i = 0; i < bvh.n_nodes; ++i) {
            const rtk::BvhNode &nd = nodes[i];
            const bool is_leaf = (nd.link & rtk::BVH_LEAF_FLAG) != 0;
            const int32_t skip = is_leaf ? i + 1 : static_cast<int32_t>(nd.link);
            if (skip <= i || skip > bvh.n_nodes) return fail(RT_ERR_DEVICE, "octant %d node %d: skip link %d", oct, i, skip);
            for (int32_t j = i + 1; j < skip; ++j)
                for (int a = 0; a < 3; ++a)
                    if (nodes[j].lo[a] < nd.lo[a] || nodes[j].hi[a] > nd.hi[a]) return fail(RT_ERR_DEVICE, "octant %d node %d is not inside its ancestor %d", oct, j, i);
            if (!is_leaf) { if (skip == i + 1) return fail(RT_ERR_DEVICE, "octant %d inner node %d has no subtree", oct, i); continue; }
            const int first = static_cast<int>(nd.link & 0x0fffffffu), count = static_cast<int>((nd.link >> 28) & 7u) + 1;
            if (count > rtk::BVH_LEAF_MAX || first + count > static_cast<int>(bvh.spheres.size())) return fail(RT_ERR_DEVICE, "octant %d leaf %d: range [%d, +%d)", oct, i, first, count);
            worst = std::max<int64_t>(worst, count);
            for (int j = first; j < first + count; ++j) {
                const rtk::BvhSphere &sp = bvh.spheres[j];
                if (sp.index < 0 || sp.index >= static_cast<int>(ns) || seen[sp.index]++) return fail(RT_ERR_DEVICE, "octant %d: sphere %d appears twice or is out of range", oct, sp.index);
                if (j > first && bvh.spheres[j - 1].index >= sp.index) return fail(RT_ERR_DEVICE, "octant %d leaf %d is not in list order", oct, i);
                const float r = std::fabs(rad[sp.index]);
                const float c[3] = {rec[4 * sp.index], rec[4 * sp.index + 1], rec[4 * sp.index + 2]};
                for (int a = 0; a < 3; ++a)
                    if (!(corner(nd.lo[a], a) < c[a] - r && c[a] + r < corner(nd.hi[a], a))) return fail(RT_ERR_DEVICE, "octant %d leaf %d does not contain sphere %d", oct, i, sp.index);
            }
        }
        for (size_t i = 0; i < ns; ++i) if (seen[i] != 1) return fail(RT_ERR_DEVICE, "octant %d: sphere %zu is in %d leaves", oct, i, seen[i]);
    }
    if (n_nodes) *n_nodes = bvh.n_nodes;
    if (max_leaf) *max_leaf = worst;
    return RT_OK;
}

int32_t rt_render_tiled(rt_v3 *image_data, int64_t width, int64_t height, const rt_camera_render_data *cam,
                        const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials,
                        const rt_options *opts) {
    rt_rect whole = {0, 0, width, height};
    return rt_render_region(image_data, width, height, whole, cam, spheres, n_spheres, materials, n_materials, opts);
}

uint32_t rt_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t draw) {
    uint32_t k0, k1;
    seed_keys(seed, k0, k1);
    const uint32_t kp = mix32(k0 ^ pixel), wp = mix32(k1 + pixel);
    const uint32_t ks = mix32(kp + 0x9e3779b9u * (sample + 1u));
    const uint32_t kb = mix32(ks ^ (0x85ebca6bu * (bounce + 1u)));
    return mix32(kb + 0xc2b2ae35u * (draw + 1u)) ^ wp;
}

} // extern "C"
