// gfx950 kernels of the path tracer: the loops of render_region (main.odin:510-591).
//
// Work decomposition
//   A wave64 works on 8x8-pixel tiles.  Every lane runs ONE flat loop whose iteration is one ray
//   segment: the closest-hit scan over all spheres, then shading.  A lane whose path ended
//   starts another sample in the same iteration, so lanes never wait for the longest path.
//   Throughput mode (QUEUE): work items are (tile, sample chunk) pairs behind one global
//   counter.  A wave takes an item, deals its units -- (pixel, sample) pairs -- to whichever lanes
//   are free, and takes the next item when the units run out, while other lanes are still finishing
//   paths of the previous one.  A finished sample's colour is added in fixed point (Q32.32), so the
//   sums are exact and independent of which lane, wave or GPU traced what: first to the wave's sums
//   for the item's tile in LDS, which go to the pixel accumulators in HBM with one set of integer
//   atomics per pixel and item (an atomic per sample and channel costs 3.5 % of the frame, and six
//   per sample -- the fixed-point mixed mode's two words per channel -- are more than the memory side
//   executes).  Lanes idle only when the whole queue is empty.
//   (Hierarchy variants: beside a tree staged in LDS there is room for the current item's sums only, and
//   samples that finish after their item add to HBM directly; the f32 kernel, which lives on occupancy
//   there, keeps its per-sample atomics.)
//   Literal modes (f64 / f32 reference order): lane l owns pixel l of one tile and one sample
//   chunk, sums its samples in order in a register and stores the sum.
//   Mixed mode: f64 paths in the reference's operation order behind the f32 filter scan, whose
//   error margin never drops a sphere the f64 test would hit (the bound is derived at the scan).  Lanes pool work as in throughput mode; every sample's colour is
//   stored to its own slot and a second kernel adds the slots in sample order, so the image is the
//   f64 literal mode's bit for bit.
//   Shading follows the reference line by line; the one loop in it that a wave would repeat until its unluckiest lane
//   is through -- the rejection sampling of random_unit_vector -- is run by the wave together (first_unit_candidate; list scan).
// Closest-hit scan (throughput and mixed mode)
//   The sphere index is wave-uniform: records of sphere PAIRS {cx0,cx1, cy0,cy1, cz0,cz1, -K0,-K1},
//   K = |c|^2 - r^2 lowered by an error margin, come through the scalar cache into SGPRs, the next group
//   of four spheres being fetched while this one is tested.  The first stage is a conservative FILTER
//   on the line of the ray, given by a unit direction e and the foot point f of the coordinate origin:
//   r^2 - dist(c, line)^2 = (c.e)^2 - K + 2 c.f - |f|^2 (main.odin:245-249 divided by a, with the
//   squares expanded and the per-ray and per-sphere parts hoisted).  Per sphere a lane spends 3 fused
//   operations on h = c.e, one on h*h - K and 3 on adding 2 c.f, all as packed instructions on two
//   spheres at once; a group of four spheres shares one compare with |f|^2.  The margins (derived at
//   the scan) guarantee that no sphere with a non-negative exact discriminant is dropped, so the image
//   is decided by the second stage alone.  Only when some lane's line meets a sphere of the group does
//   the wave leave the straight-line code; such a lane queues (lane, sphere) in a per-wave queue in LDS,
//   and the second stage -- throughput mode: the roots from the centre-to-line vector, whose rounding
//   scales with the sphere and not with its distance; mixed mode: ray_sphere_intersection in f64 as
//   written -- with the range tests of main.odin:250-261 is evaluated later, 64 entries at a
//   time by whichever lanes are at hand, each result going to its owner through an LDS atomic
//   minimum (the loop of main.odin:540-546 ends with the smallest accepted root, lowest index among
//   equals, whatever the order of evaluation).  The literal modes park one sphere per lane instead
//   and evaluate it when the lane needs the slot again, in sphere order with the shrinking t_max.
//   Throughput mode: the sphere a scattered segment starts on is tested before the scan with c = 0 (in
//   f64 the reference sees |c| ~ 1e-15 there; in f32 the rounding of c would produce spurious
//   self-intersections).  BVH variants replace the list by a per-lane, stackless traversal of a
//   bounding-volume hierarchy with the same two stages per sphere; in the mixed mode the traversal runs
//   in f32 on the rounded f64 ray with its box test widened by what that rounding can move.
//   Literal modes: records {cx, cy, cz, r*r}, the reference's expressions as written.
// Arithmetic
//   Compiled with -ffp-contract=off: the only fused operations are the explicit fma calls -- those
//   of the throughput mode's scan and roots, and, in every mode, places where the fused form is
//   exact and therefore returns what the reference's separate operations return (the draws, the
//   integer form of the f64 rejection tests).  Division and square root are IEEE correctly rounded.
//   In throughput mode a vector is divided by a scalar as one reciprocal and three products;
//   everything else follows the reference's operation order.
#include "rt_kernel.h"
#include <type_traits>

#include <algorithm>
#include <mutex>

namespace rtk {

// ----------------------------------------------------------------------------- small math
template <typename Real> struct V3 { Real x, y, z; };
template <typename Real> __device__ __forceinline__ V3<Real> operator+(V3<Real> a, V3<Real> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator-(V3<Real> a, V3<Real> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator-(V3<Real> a) { return {-a.x, -a.y, -a.z}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator*(V3<Real> a, V3<Real> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator*(Real s, V3<Real> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator*(V3<Real> a, Real s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename Real> __device__ __forceinline__ V3<Real> operator/(V3<Real> a, Real s) { return {a.x / s, a.y / s, a.z / s}; }
// dot (main.odin:28-30): left to right, no fusion
template <typename Real> __device__ __forceinline__ Real dot(V3<Real> a, V3<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); }
// IEEE maxNum: a NaN operand is ignored (v_max_f32 / v_max_f64)
__device__ __forceinline__ float max_num(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double max_num(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float min_num(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float abs_rn(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double abs_rn(double x) { return __builtin_fabs(x); }
template <typename Real> __device__ __forceinline__ Real min_std(Real a, Real b) { return (b < a) ? b : a; } // std::min
template <typename Real> __device__ __forceinline__ Real max_std(Real a, Real b) { return (a < b) ? b : a; } // std::max
// A vector divided by a scalar (main.odin:49, :200, :555): three divisions in the literal modes,
// one reciprocal and three products in the throughput arithmetic (FAST).
template <typename Real, bool FAST> __device__ __forceinline__ V3<Real> scale_down(V3<Real> v, Real s) {
    if (FAST) return v * (Real(1) / s);
    return v / s;
}
template <typename Real, bool FAST> __device__ __forceinline__ V3<Real> normalize(V3<Real> v) { return scale_down<Real, FAST>(v, sqrt_rn(dot(v, v))); } // main.odin:48-50

template <typename Real> struct Lim;
template <> struct Lim<float> { static constexpr float max_v = 3.402823466e+38f; static constexpr float eps = 1.1920928955078125e-07f; };
template <> struct Lim<double> { static constexpr double max_v = 1.7976931348623157e+308; static constexpr double eps = 2.220446049250313e-16; };

template <typename Real> struct Rec4;
template <> struct Rec4<float> { using type = float4; };
template <> struct Rec4<double> { using type = double4; };

// ----------------------------------------------------------------------------- counter RNG
// draw j (1-based) of stream (pixel, sample, bounce): mix32(kb + 0xc2b2ae35*j) ^ wp with
//   kp = mix32(k0 ^ pixel), wp = mix32(k1 + pixel), ks = mix32(kp + 0x9e3779b9*(sample+1)),
//   kb = mix32(ks ^ 0x85ebca6b*(bounce+1)).   bounce 0 = camera draws, k+1 = scatter at segment k.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
struct Draws {
    uint32_t kb, wp, c;
    __device__ __forceinline__ Draws(uint32_t ks, uint32_t wp_, uint32_t bounce) : kb(mix32(ks ^ (0x85ebca6bu * (bounce + 1u)))), wp(wp_), c(0) {}
    __device__ __forceinline__ uint32_t next_u32() { c += 0xc2b2ae35u; return mix32(kb + c) ^ wp; }
    // rand.float64(): uniform [0,1) with 24 random bits, exact in f32 and f64
    template <typename Real> __device__ __forceinline__ Real next() { return static_cast<Real>(next_u32() >> 8) * static_cast<Real>(1.0 / 16777216.0); }
    // rand.float64_range(lo, hi) is (hi-lo)*u + lo, and a result that reaches hi is pulled back below it.  For the
    // two ranges the path uses and u = k * 2^-24 this is exact arithmetic on small integers in f32 and in f64:
    //   (-1, 1 + F64_EPSILON):  hi - lo rounds to 2 (in f32 hi itself rounds to 1), 2u - 1 = (k - 2^23) * 2^-23;
    //   (-0.5, 0.5)          :  hi - lo = 1,                                        u - 0.5 = (k - 2^23) * 2^-24;
    // both representable, both below hi, so the pull-back never applies and one fused operation on the integer
    // returns the very same number.
    template <typename Real> __device__ __forceinline__ Real range_pm1() { return fma_rn(static_cast<Real>(next_u32() >> 8), static_cast<Real>(1.0 / 8388608.0), Real(-1)); }
    template <typename Real> __device__ __forceinline__ Real range_pm_half() { return fma_rn(static_cast<Real>(next_u32() >> 8), static_cast<Real>(1.0 / 16777216.0), Real(-0.5)); }
};

// The rejection loops of main.odin:196-202 and :216-222 in f64, without f64 arithmetic on rejected candidates.
// A draw is u = k * 2^-24 with a 24-bit integer k, and float64_range(-1, 1 + F64_EPSILON) maps it to
// (hi - lo) * u + lo = 2u - 1 exactly (hi - lo = 2 + 2^-52 rounds to 2; the pull-back of a value >= hi never
// applies), so a candidate coordinate is X * 2^-23 with the integer X = k - 2^23.  Its square has at most 48
// significant bits and the sum of two or three squares at most 50: dot(p, p) (main.odin:28-30) is exact in f64 and
// equals (X^2 + Y^2 + Z^2) * 2^-46.  The loop therefore tests the integers -- converted to f64, where products and
// sums below 2^53 are exact as well -- and only the accepted candidate is scaled: the same values, bit for bit.
__device__ __forceinline__ double centred_draw(Draws &g) { return static_cast<double>(static_cast<int32_t>(g.next_u32() >> 8) - 8388608); }
__device__ __forceinline__ V3<double> random_unit_vector_f64(Draws &g) {
    double X = 0.0, Y = 0.0, Z = 0.0, sq = 0.0;
    for (int it = 0; it < REJECTION_CAP; ++it) {
        X = centred_draw(g); Y = centred_draw(g); Z = centred_draw(g);
        sq = __builtin_fma(Z, Z, __builtin_fma(Y, Y, X * X)); // exact
        if (0.0 < sq && sq <= 70368744177664.0) break;         // 0 < |p|^2 <= 1 with |p|^2 = sq * 2^-46
    }
    const double scale = 1.0 / 8388608.0;
    const V3<double> p = {X * scale, Y * scale, Z * scale};
    const double len = sqrt_rn(sq * (1.0 / 70368744177664.0));
    return {p.x / len, p.y / len, p.z / len};
}

// random_unit_vector (main.odin:194-203)
template <typename Real, bool FAST> __device__ __forceinline__ V3<Real> random_unit_vector(Draws &g) {
    if constexpr (std::is_same<Real, double>::value) return random_unit_vector_f64(g);
    V3<Real> p = {Real(0), Real(0), Real(0)};
    Real lsq = Real(0);
    for (int it = 0; it < REJECTION_CAP; ++it) {
        p.x = g.range_pm1<Real>();
        p.y = g.range_pm1<Real>();
        p.z = g.range_pm1<Real>();
        lsq = dot(p, p);
        if (Real(0) < lsq && lsq <= Real(1)) break;
    }
    return scale_down<Real, FAST>(p, sqrt_rn(lsq));
}

// number of lanes below this one whose bit is set in `mask`
__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// The rejection loop of random_unit_vector run by the WAVE (main.odin:196-202; same candidates, same result).  Alone, a
// lane tries its stream's candidates one after the other -- 52 % are accepted -- and the wave repeats the loop body until
// its unluckiest lane is through: five trips where the mean lane needs 1.9.  Here every lane that needs a vector tries its
// candidate 0; then the lanes still searching (the owners) have their next candidates tried by ALL 64 lanes at once, 1 to
// 16 candidates per owner depending on how many owners are left; an owner takes the lowest-numbered accepted one, which
// is the one its own loop would have stopped at.  Two rounds in nearly every case.  What travels is the candidate's three
// 24-bit draws; the owner forms the vector from them exactly as its own loop would.
// `first_unit_candidate` must be called by all lanes of the wave that are still running, in converged control flow.
__device__ __forceinline__ uint32_t stream_draw24(uint32_t kb, uint32_t wp, uint32_t n) { return (mix32(kb + 0xc2b2ae35u * n) ^ wp) >> 8; } // draw n = 1, 2, ...
template <typename Real> __device__ __forceinline__ bool unit_candidate_accepted(uint32_t kx, uint32_t ky, uint32_t kz) {
    if constexpr (std::is_same<Real, double>::value) {
        // the integer form of random_unit_vector_f64: |p|^2 = (X^2 + Y^2 + Z^2) * 2^-46, exact
        const double X = static_cast<double>(static_cast<int32_t>(kx) - 8388608), Y = static_cast<double>(static_cast<int32_t>(ky) - 8388608),
                     Z = static_cast<double>(static_cast<int32_t>(kz) - 8388608);
        const double sq = __builtin_fma(Z, Z, __builtin_fma(Y, Y, X * X));
        return 0.0 < sq && sq <= 70368744177664.0;
    } else {
        const V3<float> p = {fma_rn(static_cast<float>(kx), 1.0f / 8388608.0f, -1.0f), fma_rn(static_cast<float>(ky), 1.0f / 8388608.0f, -1.0f),
                             fma_rn(static_cast<float>(kz), 1.0f / 8388608.0f, -1.0f)};
        const float lsq = dot(p, p);
        return 0.0f < lsq && lsq <= 1.0f;
    }
}
template <typename Real>
__device__ __forceinline__ void first_unit_candidate(const bool need, const uint32_t kb, const uint32_t wp, uint32_t *slot, const int lane,
                                                     uint32_t &kx, uint32_t &ky, uint32_t &kz) {
    bool done = !need;
    if (need) {
        kx = stream_draw24(kb, wp, 1u); ky = stream_draw24(kb, wp, 2u); kz = stream_draw24(kb, wp, 3u);
        done = unit_candidate_accepted<Real>(kx, ky, kz);
    }
    unsigned long long pending = __builtin_amdgcn_ballot_w64(!done);
    if (pending == 0ull) return;
    if (__builtin_amdgcn_ballot_w64(true) != ~0ull) {
        // some lanes of the wave have left already (the end of a launch): every lane for itself
        if (!done) {
            for (uint32_t c = 1u; c < static_cast<uint32_t>(REJECTION_CAP); ++c) {
                kx = stream_draw24(kb, wp, 3u * c + 1u); ky = stream_draw24(kb, wp, 3u * c + 2u); kz = stream_draw24(kb, wp, 3u * c + 3u);
                if (unit_candidate_accepted<Real>(kx, ky, kz)) break;
            }
        }
        return;
    }
    uint32_t base = 1u; // candidates tried so far per owner
    while (pending != 0ull) {
        const uint32_t owners = static_cast<uint32_t>(__builtin_popcountll(pending));
        const uint32_t shift = owners <= 4u ? 4u : (owners <= 8u ? 3u : (owners <= 16u ? 2u : (owners <= 32u ? 1u : 0u))); // 2^shift candidates per owner
        const uint32_t my_rank = rank_in(pending);
        if (!done) slot[my_rank] = static_cast<uint32_t>(lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t oi = static_cast<uint32_t>(lane) >> shift, c = static_cast<uint32_t>(lane) & ((1u << shift) - 1u);
        const bool helping = oi < owners;
        const int owner = helping ? static_cast<int>(slot[oi]) : lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t okb = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(owner << 2, static_cast<int>(kb)));
        const uint32_t owp = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(owner << 2, static_cast<int>(wp)));
        // the loop ends after REJECTION_CAP candidates with the last one, accepted or not
        const uint32_t cand = min(base + c, static_cast<uint32_t>(REJECTION_CAP) - 1u);
        const uint32_t hx = stream_draw24(okb, owp, 3u * cand + 1u), hy = stream_draw24(okb, owp, 3u * cand + 2u), hz = stream_draw24(okb, owp, 3u * cand + 3u);
        const bool ok = helping && (unit_candidate_accepted<Real>(hx, hy, hz) || cand == static_cast<uint32_t>(REJECTION_CAP) - 1u);
        const unsigned long long accepted = __builtin_amdgcn_ballot_w64(ok);
        // an owner's helpers are the lanes [rank << shift, (rank + 1) << shift): the lowest accepted candidate wins
        const unsigned long long mine = (accepted >> (my_rank << shift)) & ((1ull << (1u << shift)) - 1ull);
        const bool found = !done && mine != 0ull;
        const int src = found ? static_cast<int>((my_rank << shift) + static_cast<uint32_t>(__builtin_ctzll(mine))) : lane;
        const uint32_t fx = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src << 2, static_cast<int>(hx)));
        const uint32_t fy = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src << 2, static_cast<int>(hy)));
        const uint32_t fz = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src << 2, static_cast<int>(hz)));
        if (found) { kx = fx; ky = fy; kz = fz; done = true; }
        pending = __builtin_amdgcn_ballot_w64(!done);
        base += 1u << shift;
    }
}
// the unit vector of an accepted candidate, as the loops of random_unit_vector / random_unit_vector_f64 form it
template <typename Real, bool FAST> __device__ __forceinline__ V3<Real> unit_vector_of_candidate(uint32_t kx, uint32_t ky, uint32_t kz) {
    if constexpr (std::is_same<Real, double>::value) {
        const double X = static_cast<double>(static_cast<int32_t>(kx) - 8388608), Y = static_cast<double>(static_cast<int32_t>(ky) - 8388608),
                     Z = static_cast<double>(static_cast<int32_t>(kz) - 8388608);
        const double sq = __builtin_fma(Z, Z, __builtin_fma(Y, Y, X * X));
        const double scale = 1.0 / 8388608.0;
        const V3<double> p = {X * scale, Y * scale, Z * scale};
        const double len = sqrt_rn(sq * (1.0 / 70368744177664.0));
        return {p.x / len, p.y / len, p.z / len};
    } else {
        const V3<float> p = {fma_rn(static_cast<float>(kx), 1.0f / 8388608.0f, -1.0f), fma_rn(static_cast<float>(ky), 1.0f / 8388608.0f, -1.0f),
                             fma_rn(static_cast<float>(kz), 1.0f / 8388608.0f, -1.0f)};
        return scale_down<float, FAST>(p, sqrt_rn(dot(p, p)));
    }
}

// ----------------------------------------------------------------------------- optics
// reflect (main.odin:77-80)
template <typename Real> __device__ __forceinline__ V3<Real> reflect(V3<Real> v, V3<Real> n) { return v - (Real(2) * dot(v, n)) * n; }
// refract_with_relative_refractive_index, arithmetic (main.odin:168-173)
template <typename Real> __device__ __forceinline__ V3<Real> refract(V3<Real> v, V3<Real> n, Real rel) {
    Real cos_theta = min_std(dot(-v, n), Real(1));
    V3<Real> perp = rel * (v + cos_theta * n);
    V3<Real> par = (-sqrt_rn(max_std(Real(1) - dot(perp, perp), Real(0)))) * n;
    return perp + par;
}

// ----------------------------------------------------------------------------- the kernel
template <typename Real, typename Ptr> __device__ __forceinline__ V3<Real> cam3(Ptr c, int i) { return {c[3 * i], c[3 * i + 1], c[3 * i + 2]}; }
template <typename Real> constexpr size_t camera_offset() { return std::is_same<Real, double>::value ? offsetof(RenderParams, cam64) : offsetof(RenderParams, cam32); }

__device__ __forceinline__ int tile_row_of(const TileMap &m, int k) {
    return ((k / m.rows_per_strip) * m.strip_count + m.strip_index) * m.rows_per_strip + (k % m.rows_per_strip);
}

// Float to u32 as the hardware converts: truncated, NaN and negatives to 0, 2^32 and above to 2^32 - 1.  (A C++ cast leaves
// values out of range undefined, so the range tests would have to stay in front of it.)
__device__ __forceinline__ uint32_t to_u32_saturating(float x) { uint32_t r; asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t to_u32_saturating(double x) { uint32_t r; asm("v_cvt_u32_f64_e32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// Q32.32 fixed point of a non-negative f32 (truncated below 2^-32; NaN and negatives give 0, 2^32 and above 2^64 - 2^32).
// Sums of these are exact and independent of the order of addition.  The saturating conversions take care of NaN and
// negatives: hi = 0, and the low word of a NaN or negative remainder is 0 as well.
__device__ __forceinline__ unsigned long long to_q32_32(float x) {
    const uint32_t hi = to_u32_saturating(x);
    const float fr = x - static_cast<float>(hi); // exact: below 2^24 both are on x's grid, above it x is an integer
    const uint32_t lo = x >= 4294967296.0f ? 0u : to_u32_saturating(fr * 4294967296.0f);
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ unsigned long long to_q32_32(double) { return 0ull; } // never used: the f64 queue modes stage or use Q32.64
// Q32.64 fixed point of a non-negative f64 as two words: a = floor(x * 2^32) (units of 2^-32), b = the next 32 bits
// (units of 2^-64); x is truncated below 2^-64, NaN and negatives give 0, x >= 2^32 saturates.  Every step is exact:
// the differences are of numbers on a common grid and below one unit of the part already taken.  Word sums of up to
// 2^32 samples cannot overflow (a < 2^64 needs sum x < 2^32; b < 2^32 per sample).
// Saturating conversions, no range tests: a NaN or a negative x gives 0 three times (the remainders are NaN or negative too);
// for x >= 2^32 the first word saturates, x - (2^32 - 1) >= 1 makes the scaled remainder >= 2^32, and so on down.
__device__ __forceinline__ void to_q32_64(double x, unsigned long long &a, unsigned long long &b) {
    const uint32_t hi = to_u32_saturating(x);                     // truncates
    const double t1 = (x - static_cast<double>(hi)) * 4294967296.0; // in [0, 2^32)
    const uint32_t mid = to_u32_saturating(t1);
    const double t2 = (t1 - static_cast<double>(mid)) * 4294967296.0;
    a = (static_cast<unsigned long long>(hi) << 32) | mid;
    b = static_cast<unsigned long long>(to_u32_saturating(t2));
}
// The sum held by the two words, a * 2^-32 + b * 2^-64, rounded to f64 once (to nearest, ties to even): the 128-bit
// integer T = a * 2^32 + b is normalised, its top 64 bits taken with everything below them OR-ed into the lowest (far
// below the 53 bits that survive, so it only breaks ties), converted (u64 -> f64 is one correctly rounded
// operation: two exact halves and one addition) and scaled by a power of two.
__device__ __forceinline__ double q32_64_to_f64(unsigned long long a, unsigned long long b) {
    unsigned long long lo = (a << 32) + b;
    unsigned long long hi = (a >> 32) + (lo < b ? 1ull : 0ull);
    if (hi == 0ull) return static_cast<double>(lo) * 5.421010862427522e-20; // 2^-64
    const int shift = __builtin_clzll(hi); // hi < 2^34: shift >= 30
    unsigned long long top = (hi << shift) | (shift == 0 ? 0ull : (lo >> (64 - shift)));
    const unsigned long long rest = shift == 0 ? lo : (lo << shift);
    if (rest != 0ull) top |= 1ull;
    return __builtin_ldexp(static_cast<double>(top), -shift); // top * 2^(64 - shift) * 2^-64
}

typedef float F2 __attribute__((ext_vector_type(2)));
struct Pair { F2 cx, cy, cz, k; }; // two spheres' scan records, component-wise
__device__ __forceinline__ F2 fma2(F2 a, F2 b, F2 c) { return __builtin_elementwise_fma(a, b, c); } // v_pk_fma_f32


#ifndef RT_WAVES_PER_EU
// The mixed kernel is compiled for four waves per SIMD: squeezed into the 96 registers of five it spills 45 of them
// (measured, same box: 63.7 ms at five, 62.0 ms at four; without the wave queue five was 1.6 % ahead of four).  The
// other instantiations keep the compiler's choice (1..8 is no constraint).
// Through the hierarchy the mixed kernel waits on LDS or memory rather than on the vector pipe and more waves pay
// (same box, 300 spp: LDS form 46.3 ms at four, 47.0 at five -- 512-thread workgroups, so five is four --, 44.2 at
// six, 65 at eight; global form 51.0 at four and six, 46.5 at five).
// The fixed-point variants carry nine more registers (a unit's sums): four waves throughout (same box, 300 spp, LDS / global
// form: 54.3 / 60.9 ms at six / five waves, 51.4 / 58.6 at four).
#define RT_MIXED_WAVES (FIXED ? 4 : (BVH == 2 ? 6 : (BVH == 1 ? 5 : 4)))
#define RT_WAVES_PER_EU_ATTR __attribute__((amdgpu_waves_per_eu(MIXED ? RT_MIXED_WAVES : 1, MIXED ? RT_MIXED_WAVES : 8)))
#else
#define RT_WAVES_PER_EU_ATTR __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU)))
#endif

// BVH: 0 = list scan; 1 = hierarchy in global memory (16-byte nodes, one order per direction octant);
//      2 = hierarchy in LDS (boxes once, 16-bit links per octant), for hierarchies small enough
// MIXED: f64 paths in the reference's literal operation order (Real = double, FMA = false) whose
//        closest-hit scan is preceded by the packed-f32 filter of the throughput mode, made
//        conservative: a sphere is dropped only when its discriminant is negative by more than every
//        rounding error the f32 evaluation can have made.  Samples are dealt to free lanes (QUEUE)
//        and every finished sample colour is stored, so that a second kernel can add them in
//        sample order (main.odin:575).
// FIXED (with MIXED): a finished sample's colour is added to its pixel's Q32.64 accumulators with integer atomics
//        (exact, order-independent) instead of being staged for the ordered sum.
template <typename Real, bool FMA, int FETCH, bool QUEUE, int BVH = 0, bool MIXED = false, bool FIXED = false>
__global__ __launch_bounds__(BVH == 2 ? BVH_LDS_BLOCK_THREADS : BLOCK_THREADS) RT_WAVES_PER_EU_ATTR void render_kernel(const RenderParams P) {
    static_assert(BVH == 0 || ((FMA || MIXED) && QUEUE && FETCH == FETCH_SCALAR), "the hierarchy is traversed in f32, by the throughput and the mixed mode");
    static_assert(!MIXED || (std::is_same<Real, double>::value && !FMA && QUEUE && FETCH == FETCH_SCALAR), "the mixed mode is f64 literal arithmetic behind the scalar-fetch f32 filter");
    static_assert(!FIXED || MIXED, "fixed-point sums of f64 colours belong to the mixed mode");
    constexpr bool FILTER = FMA || MIXED; // the scan is the packed-f32 candidate filter
    // Candidates of the filter go to a queue per wave in LDS and are evaluated 64 at a time by whichever lanes there
    // are, instead of each lane evaluating its own as they come (which ran the evaluation code about four times per
    // segment with a quarter of the lanes each time).  See `enqueue` / `flush` in the scan.
#ifdef RT_NO_WAVE_QUEUE
    constexpr bool WQ = false;
#else
    constexpr bool WQ = FILTER && BVH == 0 && FETCH == FETCH_SCALAR;
#endif
    __shared__ uint32_t wq_entry[WQ ? WAVES_PER_BLOCK : 1][WQ ? WQ_CAP : 1];                    // (owner lane << 24) | sphere
    __shared__ unsigned long long wq_best[WQ ? WAVES_PER_BLOCK : 1][WQ ? 64 : 1];             // per owner lane: smallest key so far
    __shared__ unsigned long long wq_val[(WQ && MIXED) ? WAVES_PER_BLOCK : 1][(WQ && MIXED) ? WQ_CAP : 1]; // mixed: the entries' roots
    __shared__ uint32_t wq_besti[(WQ && MIXED) ? WAVES_PER_BLOCK : 1][(WQ && MIXED) ? 64 : 1];  // mixed: lowest sphere index among equal roots
    // Tile sums (queue modes that add fixed-point colours): the colours of finished samples are added to
    // per-wave accumulators in LDS -- one set for the tile of the wave's current work item and one for the previous
    // item's, since its last paths are still in flight -- and a tile's sums go to the accumulators in HBM with one set
    // of integer atomics per pixel when the wave takes the next item but one.  Integer sums: exact whatever the grouping.
    // Per sample that is SUM_WORDS LDS atomics and SUM_WORDS / (samples per item) global ones; a sample that outlives two
    // item changes (a few per cent: paths of ten and more segments) adds to HBM directly.
    // Hierarchy variants: where the tree is staged into LDS (BVH == 2) there is room for the current item's set only (its
    // stragglers add to HBM directly: about a third of the samples, 2.5 atomics per sample and channel pair in all); the f32
    // kernel, which runs six waves per SIMD there, keeps its per-sample atomics instead of giving two of them up for the sums.
    constexpr bool LSUM = QUEUE && FETCH == FETCH_SCALAR && ((FMA && BVH != 2) || (MIXED && FIXED));
    static_assert(!(MIXED && FIXED) || LSUM, "the fixed-point mixed mode adds its colours through the tile sums");
    constexpr int GENS = BVH == 2 ? 1 : 2; // sets of sums: items of the wave that are current at a time
    constexpr int SUM_WORDS = (MIXED && FIXED) ? 6 : 3; // 64-bit words per pixel: Q32.32 per channel, or Q32.64 as two words
    __shared__ unsigned long long tile_sum[LSUM ? (BVH == 2 ? BVH_LDS_BLOCK_THREADS / 64 : WAVES_PER_BLOCK) : 1][LSUM ? GENS : 1][LSUM ? 64 * SUM_WORDS : 1];
    // queue modes: the wave runs random_unit_vector's rejection loop together (first_unit_candidate); 64 words of LDS per wave
    // (kernels with the wave candidate queue lend it the queue's entries, which are free while a segment is shaded).
    // List scan only: in the hierarchy variants, which live on occupancy, its registers cost more than it saves (measured
    // same-box: f32 from LDS 33.0 -> 39.8 ms as the kernel drops from six waves per SIMD to four, mixed +2 %).
    constexpr bool COOP = QUEUE && BVH == 0;
    __shared__ uint32_t coop_slot[(COOP && !WQ) ? (BVH == 2 ? BVH_LDS_BLOCK_THREADS / 64 : WAVES_PER_BLOCK) : 1][(COOP && !WQ) ? 64 : 1];
    const int wave_in_block = threadIdx.x >> 6;
    if constexpr (LSUM) {
        for (int k = threadIdx.x & 63; k < GENS * 64 * SUM_WORDS; k += 64) (&tile_sum[wave_in_block][0][0])[k] = 0ull;
    }
    if constexpr (WQ) {
        wq_best[wave_in_block][threadIdx.x & 63] = ~0ull;
        if constexpr (MIXED) wq_besti[wave_in_block][threadIdx.x & 63] = ~0u;
    }
    using R4 = typename Rec4<Real>::type;
    const R4 *__restrict__ g_rec = static_cast<const R4 *>(P.sphere_rec);
    const int n = P.n_spheres; // padded count, a multiple of SCAN_STEP

    // The records never change during a launch.  Reading them through the constant address
    // space tells the compiler so: a wave-uniform index then becomes a scalar-cache load into
    // SGPRs even though the kernel also writes global memory (its atomics and stores would
    // otherwise make every such load a per-lane vector load).
    typedef const Real __attribute__((address_space(4))) *ConstReal;
    ConstReal c_rec = (ConstReal)(P.sphere_rec);
    extern __shared__ double4 lds_raw[];
    R4 *l_rec = reinterpret_cast<R4 *>(lds_raw);
    if (FETCH == FETCH_LDS) {
        for (int i = threadIdx.x; i < n + GROUP; i += BLOCK_THREADS) l_rec[i] = g_rec[i];
        __syncthreads();
    }
    // hierarchy in LDS: n boxes of 16 bytes, then 8 * n links
    uint4 *l_boxes = reinterpret_cast<uint4 *>(lds_raw);
    uint32_t *l_links = reinterpret_cast<uint32_t *>(l_boxes + P.bvh_n_nodes);
    if constexpr (BVH == 2) {
        const uint4 *gb = static_cast<const uint4 *>(P.bvh_lds_boxes);
        const uint32_t *gl = static_cast<const uint32_t *>(P.bvh_lds_links);
        for (int i = threadIdx.x; i < P.bvh_n_nodes; i += BVH_LDS_BLOCK_THREADS) l_boxes[i] = gb[i];
        for (int i = threadIdx.x; i < 8 * P.bvh_n_nodes; i += BVH_LDS_BLOCK_THREADS) l_links[i] = gl[i];
        __syncthreads();
    }
    // filter modes: pair j = spheres 2j and 2j+1 as {cx0,cx1, cy0,cy1, cz0,cz1, -K'0,-K'1}
    auto pair = [&](int j) -> Pair {
        Pair p;
        if constexpr (FETCH == FETCH_LDS) {
            const float *l = reinterpret_cast<const float *>(lds_raw) + 8 * j;
            p.cx = F2{l[0], l[1]}; p.cy = F2{l[2], l[3]}; p.cz = F2{l[4], l[5]}; p.k = F2{l[6], l[7]};
        } else {
            typedef const float __attribute__((address_space(4))) *ConstF;
            ConstF c = (ConstF)(P.sphere_rec) + 8 * j;
            p.cx = F2{c[0], c[1]}; p.cy = F2{c[2], c[3]}; p.cz = F2{c[4], c[5]}; p.k = F2{c[6], c[7]};
        }
        return p;
    };
    auto rec = [&](int i) -> R4 {
        if constexpr (FETCH == FETCH_LDS) return l_rec[i];
        else { R4 r; r.x = c_rec[4 * i]; r.y = c_rec[4 * i + 1]; r.z = c_rec[4 * i + 2]; r.w = c_rec[4 * i + 3]; return r; }
    };

    const int lane = threadIdx.x & 63;

    // ---- the work a lane is on: pixel (u, v), sample s (index within the call)
    int u = 0, v = 0;
    uint32_t s = 0, pixel = 0, kp = 0, wp = 0, ks = 0;
    // literal modes: one tile and one sample chunk per wave
    uint32_t s_end = 0;
    bool inside = false;
    // A unit of work in queue mode: one (pixel, sample) pair.  (Measured on the headline frame: units of one pixel's whole
    // sample chunk, summed in registers, cost 2-5 % where the atomics are not the bound: a pixel's paths are alike, so
    // the lanes that drew the long ones fall behind the others and the wave's iterations are worse filled.  The tile
    // sums above take the atomics away without that cost.)
    // queue mode, all wave-uniform: the current item and its units
    uint32_t unit_next = 0, unit_total = 0, npx = 0, item_s_begin = 0, item_s_end = 0;
    int x0 = 0, y0 = 0, tw = TILE, item_tile = 0;
    bool queue_open = true;
    // queue mode, per lane: the lane has no unit (and asks for one)
    bool need_unit = QUEUE;
    // tile sums: the generation of the wave's current item (it counts the items the wave has taken; wave-uniform), the items
    // whose sums the two LDS sets hold (~0: none), and per lane (generation << 6 | pixel of the tile) of the sample it traces
    uint32_t cur_gen = 0, gen_item0 = ~0u, gen_item1 = ~0u, my_slot = 0;
    // the sums of one set go to HBM and the set is cleared (wave-uniform call)
    auto flush_tile = [&](const int set, const uint32_t item) {
        if constexpr (LSUM) {
            if (item == ~0u) return;
            const int tile = static_cast<int>(item % static_cast<uint32_t>(P.map.n_tiles));
            const int fx0 = P.min_x + (tile % P.map.tiles_x) * TILE;
            const int fy0 = P.min_y + tile_row_of(P.map, tile / P.map.tiles_x) * TILE;
            const int ftw = min(TILE, P.max_x - fx0);
            const uint32_t fnpx = static_cast<uint32_t>(ftw * min(TILE, P.max_y - fy0));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const unsigned long long here = __builtin_amdgcn_ballot_w64(true);
            const uint32_t n_here = static_cast<uint32_t>(__builtin_popcountll(here));
            for (uint32_t q = rank_in(here); q < fnpx; q += n_here) {
                const int fu = fx0 + static_cast<int>(q % static_cast<uint32_t>(ftw)), fv = fy0 + static_cast<int>(q / static_cast<uint32_t>(ftw));
                unsigned long long *src = &tile_sum[wave_in_block][set][q * SUM_WORDS];
                unsigned long long *dst = static_cast<unsigned long long *>(P.accum) + (static_cast<int64_t>(fu) + static_cast<int64_t>(fv) * P.width) * SUM_WORDS;
#pragma unroll
                for (int c = 0; c < SUM_WORDS; ++c) {
                    const unsigned long long w = src[c];
                    src[c] = 0ull;
                    if (w != 0ull) atomicAdd(dst + c, w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else { (void)set; (void)item; }
    };
    // ordered mixed mode: where this lane's sample colours are stored, (tile * 64 + pixel of the tile) * 3 within a sample's slab
    int64_t stage_px = 0;

    if constexpr (!QUEUE) {
        const int tile = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
        if (tile >= P.map.n_tiles) return; // whole wave
        u = P.min_x + (tile % P.map.tiles_x) * TILE + (lane & 7);
        v = P.min_y + tile_row_of(P.map, tile / P.map.tiles_x) * TILE + (lane >> 3);
        inside = (u < P.max_x) && (v < P.max_y);
        s = min(P.sample_count, blockIdx.y * P.chunk_samples);
        s_end = min(P.sample_count, s + P.chunk_samples);
        pixel = static_cast<uint32_t>(u) + static_cast<uint32_t>(v) * static_cast<uint32_t>(P.width);
        kp = mix32(P.k0 ^ pixel);
        wp = mix32(P.k1 + pixel);
    }

    V3<Real> ro = {Real(0), Real(0), Real(0)}, rd = {Real(0), Real(0), Real(0)}; // ray
    V3<Real> color = {Real(1), Real(1), Real(1)};                                // sample_color
    V3<Real> acc = {Real(0), Real(0), Real(0)};                                  // pixel_color (literal modes)
    int bounce = 0;
    bool fresh = !QUEUE; // the lane starts a camera ray in this iteration
    // throughput arithmetic: the sphere the current segment starts on (-1 for a camera ray) and its centre
    int start_sphere = -1;
    V3<Real> start_center = {Real(0), Real(0), Real(0)};
    // mixed mode: bound on |c| of main.odin:248 for the sphere this segment starts on (see the scan), and the
    // reach + |o| of the segment being traced, from which the next segment's bound is formed
    float start_cc_bound = 0.0f, reach_o = 0.0f;
    int skip_sphere = -1; // mixed mode: the sphere this segment starts on when it may be dropped on the sign of h alone, else -1
    uint32_t segments = 0, started = 0;
    uint32_t bvh_nodes = 0, bvh_tests = 0; // hierarchy mode: boxes and spheres this lane tested
#ifdef RT_DIAG_BVH
    uint32_t bvh_wave_steps = 0; // diagnostic build: trips of the traversal loop counted by the lowest active lane
#endif
#ifdef RT_DIAG_PHASES
    // diagnostic build: wave cycles (s_memtime) per phase of the segment loop, summed over waves into stats[5..7] and
    // [1..2]: 5 = dealing + camera ray, 6 = closest-hit scan without its slow path, 7 = slow path (park + resolve),
    // 1 = shading of a hit, 2 = background + end of sample
    unsigned long long ph_fresh = 0, ph_scan = 0, ph_slow = 0, ph_hit = 0, ph_end = 0, ph_t = 0, ph_in = 0;
#ifdef RT_DIAG_FLUSH
    unsigned long long ph_flush = 0; // with RT_DIAG_PHASES: the evaluation of the wave queue alone, reported in place of the shading time
#endif
#define RT_PHASE(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (acc) += now_ - ph_t; ph_t = now_; } while (0)
#else
#define RT_PHASE(acc) ((void)0)
#endif
#ifdef RT_DIAG_COUNTERS
    // diagnostic build: how often the scan leaves its straight-line code
    uint32_t diag_entries = 0, diag_resolves = 0, diag_parked = 0;
#define RT_DIAG_COUNT(x) (++(x))
#else
#define RT_DIAG_COUNT(x) ((void)0)
#endif

#ifdef RT_DIAG_PHASES
    ph_t = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        if constexpr (QUEUE) {
            // Deal units to the lanes that need one, in lane order, taking new items from the global
            // queue as the current one runs out.  `pending` and all queue state are wave-uniform, so
            // this loop does not diverge.
            unsigned long long pending = __builtin_amdgcn_ballot_w64(need_unit);
            bool got = false;
            while (pending != 0ull) {
                if (unit_next >= unit_total) {
                    if (!queue_open) break;
                    uint32_t item = 0u;
                    if (rank_in(__builtin_amdgcn_ballot_w64(true)) == 0u) item = atomicAdd(P.queue, 1u); // one lane per wave
                    item = __builtin_amdgcn_readfirstlane(item);
                    if (item >= P.n_items) {
                        queue_open = false;
                        if constexpr (LSUM) {
                            // no more items: both sets go out now, and what is still in flight adds to HBM directly
                            flush_tile(0, gen_item0);
                            if constexpr (GENS == 2) flush_tile(1, gen_item1);
                            gen_item0 = ~0u; gen_item1 = ~0u;
                            cur_gen += static_cast<uint32_t>(GENS);
                        }
                        break;
                    }
                    if constexpr (LSUM) {
                        // the new item takes over the set of the item before the previous one (with one set: of the previous
                        // one), whose sums go out first
                        ++cur_gen;
                        if (GENS == 2 && (cur_gen & 1u)) { flush_tile(1, gen_item1); gen_item1 = item; } else { flush_tile(0, gen_item0); gen_item0 = item; }
                    }
                    const uint32_t chunk = item / static_cast<uint32_t>(P.map.n_tiles);
                    const int tile = static_cast<int>(item % static_cast<uint32_t>(P.map.n_tiles));
                    item_tile = tile;
                    x0 = P.min_x + (tile % P.map.tiles_x) * TILE;
                    y0 = P.min_y + tile_row_of(P.map, tile / P.map.tiles_x) * TILE;
                    tw = min(TILE, P.max_x - x0);
                    item_s_begin = min(P.sample_count, chunk * P.chunk_samples);
                    item_s_end = min(P.sample_count, item_s_begin + P.chunk_samples);
                    npx = static_cast<uint32_t>(tw * min(TILE, P.max_y - y0)); // pixels of the tile inside the region
                    unit_total = npx * (item_s_end - item_s_begin);
                    unit_next = 0u;
                    continue;
                }
                const uint32_t avail = unit_total - unit_next;
                const bool take = need_unit && !got && rank_in(pending) < avail;
                if (take) {
                    const uint32_t index = unit_next + rank_in(pending);
                    uint32_t q; // pixel within the tile
                    uint32_t si; // sample within the item
                    if (npx == 64u) { q = index & 63u; si = index >> 6; } else { q = index % npx; si = index / npx; }
                    s = item_s_begin + si;
                    if constexpr (LSUM) my_slot = (cur_gen << 6) | q;
                    if (tw == TILE) { u = x0 + static_cast<int>(q & 7u); v = y0 + static_cast<int>(q >> 3); }
                    else { u = x0 + static_cast<int>(q % static_cast<uint32_t>(tw)); v = y0 + static_cast<int>(q / static_cast<uint32_t>(tw)); }
                    pixel = static_cast<uint32_t>(u) + static_cast<uint32_t>(v) * static_cast<uint32_t>(P.width);
                    kp = mix32(P.k0 ^ pixel);
                    wp = mix32(P.k1 + pixel);
                    if constexpr (MIXED && !FIXED)
                        stage_px = (static_cast<int64_t>(item_tile) * 64 + (v - y0) * TILE + (u - x0)) * 3;
                    got = true;
                }
                const unsigned long long taken = __builtin_amdgcn_ballot_w64(take);
                unit_next += static_cast<uint32_t>(__builtin_popcountll(taken));
                pending &= ~taken;
            }
            if (need_unit) {
                if (!got) break; // the queue is empty: this lane is done
                need_unit = false;
                fresh = true;
            }
        } else {
            if (fresh && !(inside && s < s_end)) break;
        }
        if (fresh) {
            // camera ray of global sample sample_begin + s (main.odin:519-534)
            ++started;
            ks = mix32(kp + 0x9e3779b9u * (P.sample_begin + s + 1u));
            Draws g(ks, wp, 0u);
            Real jx = g.range_pm_half<Real>();
            Real jy = g.range_pm_half<Real>();
            // The 19 camera constants are read here, through the scalar cache from the kernel's argument segment, every
            // time a wave starts camera rays: kept in scalar registers across the scan they are spilled to vector
            // lanes and read back one by one (mixed mode: 1 % of the frame).  The opaque zero keeps the loads in the loop
            // (list scan; through the hierarchy, with fewer scalar registers in use, the compiler's placement is as good).
            typedef const Real __attribute__((address_space(4))) *ConstCam;
            int opaque_zero = 0;
            if constexpr (BVH == 0) asm volatile("" : "+s"(opaque_zero));
            ConstCam C = (ConstCam)((const char __attribute__((address_space(4))) *)(__builtin_amdgcn_kernarg_segment_ptr()) + camera_offset<Real>()) + opaque_zero;
            V3<Real> psp = cam3<Real>(C, 3) + (static_cast<Real>(u) + jx) * cam3<Real>(C, 1) + (static_cast<Real>(v) + jy) * cam3<Real>(C, 2);
            ro = cam3<Real>(C, 0);
            if (C[18] > Real(0)) {
                // random_point_on_disk (main.odin:214-223)
                Real px = Real(0), py = Real(0);
                if constexpr (std::is_same<Real, double>::value) {
                    // the integer form of the loop (see random_unit_vector_f64): |p|^2 = (X^2 + Y^2) * 2^-46, exact
                    for (int it = 0; it < REJECTION_CAP; ++it) {
                        px = centred_draw(g); py = centred_draw(g);
                        if (__builtin_fma(py, py, px * px) <= 70368744177664.0) break;
                    }
                    px = px * (1.0 / 8388608.0); py = py * (1.0 / 8388608.0);
                } else
                for (int it = 0; it < REJECTION_CAP; ++it) {
                    px = g.range_pm1<Real>();
                    py = g.range_pm1<Real>();
                    if (px * px + py * py <= Real(1)) break;
                }
                ro = ro + ((px * cam3<Real>(C, 4)) + (py * cam3<Real>(C, 5)));
            }
            rd = psp - ro;
            color = {Real(1), Real(1), Real(1)};
            bounce = 0;
            start_sphere = -1;
            fresh = false;
        }

        RT_PHASE(ph_fresh);
        // ---------------- closest-hit scan (main.odin:536-546 with :240-263 inlined)
        const Real a = dot(rd, rd);
        const Real inv_a = FMA ? Real(1) / a : Real(0); // throughput arithmetic: roots are (h -+ sqrt(disc)) * (1/a)
        const Real t_min = static_cast<Real>(0.001); // SHADOW_ACNE_RAY_OFFSET
        Real closest = Lim<Real>::max_v;
        int hit = -1;
        Real pend_h = Real(0), pend_disc = Real(0);
        int pend = -1;

        auto resolve = [&]() {
#ifdef RT_DIAG_COUNTERS
            if (rank_in(__builtin_amdgcn_ballot_w64(true)) == 0u) ++diag_resolves; // one lane per execution by the wave
#endif
            if constexpr (FMA) {
                // Throughput arithmetic: the scan's filter value is a difference of numbers of the size of
                // |centre|^2 and only selects candidates.  A candidate's roots
                // come from the vector l from the centre to the line's closest point, whose length
                // is at most of the size of the radius:
                //     oc = centre - o,  q = (d.oc)/a,  l = oc - q*d,  D = r^2 - |l|^2,
                //     roots q -+ sqrt(D/a)            (main.odin:249-258 with disc = a*D),
                // so their rounding error scales with the sphere, not with its distance.
                const R4 sp = static_cast<const R4 *>(P.sphere_center)[pend]; // {centre, r*r}
                const Real ocx = sp.x - ro.x, ocy = sp.y - ro.y, ocz = sp.z - ro.z;
                const Real q = fma_rn(ocz, rd.z, fma_rn(ocy, rd.y, ocx * rd.x)) * inv_a;
                const Real lx = fma_rn(-q, rd.x, ocx), ly = fma_rn(-q, rd.y, ocy), lz = fma_rn(-q, rd.z, ocz);
                const Real D = sp.w - fma_rn(lz, lz, fma_rn(ly, ly, lx * lx));
                if (D >= Real(0)) {
                    const Real sq = sqrt_rn(D * inv_a);
                    Real root = q - sq;
                    bool rejected = (root < t_min) || (closest <= root);
                    if (rejected) {
                        root = q + sq;
                        rejected = (root < t_min) || (closest <= root);
                    }
                    if (!rejected && root < closest) { closest = root; hit = pend; }
                }
            } else if constexpr (MIXED) {
                // ray_sphere_intersection as written (main.odin:245-261) in f64 for the parked
                // candidate, then the caller's `t < closest_t` (:542-545).  The filter only decided
                // that this sphere has to be looked at; nothing it computed is used here.
                const R4 sp = static_cast<const R4 *>(P.sphere_center)[pend]; // {centre, r*r}
                const Real ocx = sp.x - ro.x, ocy = sp.y - ro.y, ocz = sp.z - ro.z;
                const Real h = rd.x * ocx + rd.y * ocy + rd.z * ocz;
                const Real cc = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w;
                const Real disc = h * h - a * cc;
                if (!(disc < Real(0))) {
                    const Real sq = sqrt_rn(disc);
                    Real root = (h - sq) / a;
                    bool rejected = (root < t_min) || (closest <= root);
                    if (rejected) {
                        root = (h + sq) / a;
                        rejected = (root < t_min) || (closest <= root);
                    }
                    if (!rejected && root < closest) { closest = root; hit = pend; }
                }
            } else {
                // main.odin:252-261 for the parked candidate, then :542-545
                Real sq = sqrt_rn(pend_disc);
                Real root = (pend_h - sq) / a;
                bool rejected = (root < t_min) || (closest <= root);
                if (rejected) {
                    root = (pend_h + sq) / a;
                    rejected = (root < t_min) || (closest <= root);
                }
                if (!rejected && root < closest) { closest = root; hit = pend; }
            }
            pend = -1;
        };
        // ---- wave queue (WQ): the order in which a lane's candidates are evaluated does not matter.  For one sphere
        // ray_sphere_intersection (main.odin:250-261) returns v = the near root if it is >= t_min, else the far root if
        // that is >= t_min -- provided v < t_max; and since t_max only ever shrinks to the smallest v seen, the loop of
        // main.odin:540-546 ends with the smallest v over all spheres, the lowest index among equal ones (strict `<`).
        // So every candidate is evaluated on its own, by any lane, and the results are combined per owner with a minimum.
        uint32_t qcount = 0; // entries queued by this wave (wave-uniform)
        auto bperm = [&](const int owner, const float x) -> float { return __int_as_float(__builtin_amdgcn_ds_bpermute(owner << 2, __float_as_int(x))); };
        auto bperm_d = [&](const int owner, const double x) -> double {
            const int lo = __builtin_amdgcn_ds_bpermute(owner << 2, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(owner << 2, __double2hiint(x));
            return __hiloint2double(hi, lo);
        };
        auto flush = [&]() {
            if constexpr (WQ) {
#ifdef RT_DIAG_FLUSH
                const unsigned long long flush_t0 = __builtin_amdgcn_s_memtime();
#endif
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // lanes whose work ran out have left the loop: the entries are dealt to the lanes still here
                const unsigned long long workers = __builtin_amdgcn_ballot_w64(true);
                const uint32_t n_workers = static_cast<uint32_t>(__builtin_popcountll(workers)), me = rank_in(workers);
                for (uint32_t base = 0; base < qcount; base += n_workers) {
#ifdef RT_DIAG_COUNTERS
                    if (me == 0u) ++diag_resolves; // one lane per execution by the wave
#endif
                    const bool have = base + me < qcount;
                    const uint32_t e = have ? wq_entry[wave_in_block][base + me] : 0u;
                    const int owner = static_cast<int>(e >> 24), sph = static_cast<int>(e & 0xffffffu);
                    const R4 sp = static_cast<const R4 *>(P.sphere_center)[sph]; // {centre, r*r}
                    if constexpr (FMA) {
                        // the owner's ray, then the roots from the centre-to-line vector as in `resolve`
                        const float ox = bperm(owner, ro.x), oy = bperm(owner, ro.y), oz = bperm(owner, ro.z);
                        const float dx = bperm(owner, rd.x), dy = bperm(owner, rd.y), dz = bperm(owner, rd.z), ia = bperm(owner, inv_a);
                        const float ocx = sp.x - ox, ocy = sp.y - oy, ocz = sp.z - oz;
                        const float q = fma_rn(ocz, dz, fma_rn(ocy, dy, ocx * dx)) * ia;
                        const float lx = fma_rn(-q, dx, ocx), ly = fma_rn(-q, dy, ocy), lz = fma_rn(-q, dz, ocz);
                        const float D = sp.w - fma_rn(lz, lz, fma_rn(ly, ly, lx * lx));
                        if (have && D >= 0.0f) {
                            const float sq = sqrt_rn(D * ia);
                            float v = q - sq;
                            if (!(v >= t_min)) v = q + sq;
                            if (v >= t_min) atomicMin(&wq_best[wave_in_block][owner], (static_cast<unsigned long long>(__float_as_uint(v)) << 32) | static_cast<unsigned long long>(sph));
                        }
                    } else {
                        // ray_sphere_intersection as written (main.odin:245-261) in f64 on the owner's ray
                        const double ox = bperm_d(owner, ro.x), oy = bperm_d(owner, ro.y), oz = bperm_d(owner, ro.z);
                        const double dx = bperm_d(owner, rd.x), dy = bperm_d(owner, rd.y), dz = bperm_d(owner, rd.z), oa = bperm_d(owner, a);
                        const double ocx = sp.x - ox, ocy = sp.y - oy, ocz = sp.z - oz;
                        const double h = dx * ocx + dy * ocy + dz * ocz;
                        const double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w;
                        const double disc = h * h - oa * cc;
                        unsigned long long vbits = ~0ull;
                        if (have && !(disc < 0.0)) {
                            const double sq = sqrt_rn(disc);
                            double v = (h - sq) / oa;
                            if (!(v >= t_min)) v = (h + sq) / oa;
                            if (v >= t_min) vbits = static_cast<unsigned long long>(__double_as_longlong(v));
                        }
                        if (have) {
                            wq_val[wave_in_block][base + me] = vbits;
                            if (vbits != ~0ull) atomicMin(&wq_best[wave_in_block][owner], vbits);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if constexpr (MIXED) {
                    // among the entries that reached their owner's smallest root, the lowest sphere index
                    for (uint32_t base = 0; base < qcount; base += n_workers) {
                        if (base + me < qcount) {
                            const uint32_t e = wq_entry[wave_in_block][base + me];
                            const unsigned long long vbits = wq_val[wave_in_block][base + me];
                            if (vbits != ~0ull && vbits == wq_best[wave_in_block][e >> 24]) atomicMin(&wq_besti[wave_in_block][e >> 24], e & 0xffffffu);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const unsigned long long vbits = wq_best[wave_in_block][lane];
                    const uint32_t bi = wq_besti[wave_in_block][lane];
                    wq_best[wave_in_block][lane] = ~0ull;
                    wq_besti[wave_in_block][lane] = ~0u;
                    if (vbits != ~0ull) {
                        const Real v = static_cast<Real>(__longlong_as_double(static_cast<long long>(vbits)));
                        if (v < closest) { closest = v; hit = static_cast<int>(bi); }
                    }
                } else {
                    const unsigned long long key = wq_best[wave_in_block][lane];
                    wq_best[wave_in_block][lane] = ~0ull;
                    if (key != ~0ull) {
                        const Real v = static_cast<Real>(__uint_as_float(static_cast<uint32_t>(key >> 32)));
                        if (v < closest) { closest = v; hit = static_cast<int>(key & 0xffffffu); }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                qcount = 0u;
#ifdef RT_DIAG_FLUSH
                ph_flush += __builtin_amdgcn_s_memtime() - flush_t0;
#endif
            }
        };
        // the uncommon continuation for one sphere: some lane's line meets it
        // filter modes, per-ray values of the filter (set below): the origin's parameter on the filter's line, the lowered
        // threshold, and the bound below which the centre's parameter relative to the origin is certainly negative
        float s_o = 0.0f, thr_l = 0.0f, behind = 0.0f;
        skip_sphere = -1;
        auto park = [&](const auto h, const auto cc, const auto disc, const int i) {
            using T = decltype(disc);
            // filter modes: `disc` is the filter's value w, which carries the error margin (see below); a sphere is kept
            // unless w is below the threshold, and a NaN keeps it
            bool meets;
            if constexpr (FILTER) meets = !(disc < thr_l); else meets = (disc >= T(0));
            const unsigned long long meets_mask = __builtin_amdgcn_ballot_w64(meets);
            if (meets_mask != 0ull) {
                // A line that meets the sphere only behind the origin (h <= 0) while the origin
                // is outside it (cc >= 0) has both roots <= 0: never a hit, no need to park it.
                // (the sphere the segment starts on was tested before the scan)
                // Filter modes: only when both signs are certain despite the f32 rounding -- the centre's parameter
                // relative to the origin, hs, is below -behind, and hs^2 exceeds the lowered discriminant w - thr, which
                // makes c = hs^2 - disc of main.odin:248 positive (per unit direction).
                // (plain & and | on the lane predicates: each is a compare into a scalar mask and the combination scalar
                // logic, where && and || would branch around the second compare)
                bool ahead;
#ifdef RT_ENQUEUE_ALL
                constexpr bool ENQUEUE_ALL = WQ; // measurement build: the mixed modes as well (+1.9 % there: the f64 evaluation of the extra entries costs more)
#else
                constexpr bool ENQUEUE_ALL = WQ && FMA;
#endif
                // Throughput mode: every lane whose line meets the sphere queues it, and the evaluation -- cheap in f32, and run 64
                // wide -- rejects what lies behind the origin (its roots are below t_min); the nine instructions of the test below
                // for a handful of lanes cost more (measured: -0.7 %).
                if constexpr (ENQUEUE_ALL) { (void)cc; (void)h; ahead = true; } else
                if constexpr (FILTER) {
                    (void)cc;
                    const float hs = h - s_o, dl = disc - thr_l;
                    ahead = !((hs < -behind) & ((fma_rn(hs, hs, -dl) > 0.0f) | (i == skip_sphere)));
                } else ahead = (h > T(0)) | (cc < T(0));
                const bool cand = meets & ahead & (!FMA | (i != start_sphere));
                if constexpr (WQ) {
                    // Throughput mode: the mask of `cand` from the masks of its compares by scalar logic -- a ballot of the
                    // combined predicate is lowered to a select and a second compare on the vector side (-0.2 ... -0.5 %, at the
                    // noise of a same-box comparison; in the mixed modes the same change cost 0.6 %)
                    unsigned long long cmask;
                    if constexpr (FMA && ENQUEUE_ALL) cmask = meets_mask & __builtin_amdgcn_ballot_w64(i != start_sphere);
                    else cmask = __builtin_amdgcn_ballot_w64(cand);
                    if (cmask != 0ull) {
                        const uint32_t n_new = static_cast<uint32_t>(__builtin_popcountll(cmask));
                        if (qcount + n_new > static_cast<uint32_t>(WQ_CAP)) flush();
                        if (cand) {
                            wq_entry[wave_in_block][qcount + rank_in(cmask)] = (static_cast<uint32_t>(lane) << 24) | static_cast<uint32_t>(i);
                            RT_DIAG_COUNT(diag_parked);
                        }
                        qcount += n_new;
                    }
                    return;
                }
                if (__builtin_amdgcn_ballot_w64(cand && pend >= 0) != 0ull) {
                    if (pend >= 0) resolve();
                }
                if (cand) {
                    if constexpr (!FILTER) { pend_h = h; pend_disc = disc; }
                    pend = i;
                    RT_DIAG_COUNT(diag_parked);
                }
            }
        };
        // The record array is padded to a multiple of SCAN_STEP with records that can never be
        // hit, plus one more group, so the next group is always loadable while the current one
        // is tested.  One compare covers a group of four spheres: the largest of the four
        // discriminants (NaNs ignored) is negative in every lane for most groups.
        if constexpr (FILTER) {
            // The closest-hit scan of the throughput and the mixed mode is a FILTER in seven fused f32 operations per
            // sphere that must never drop a sphere whose exact discriminant (main.odin:249) is non-negative; what it
            // keeps is decided by the second stage (throughput mode: the roots from the centre-to-line vector; mixed
            // mode: ray_sphere_intersection as written, in f64), so the image does not depend on the filter's rounding.
            // The ray (mixed mode: the f64 ray rounded to f32) becomes a unit direction e and the foot point f of the
            // coordinate origin on its line (f.e = 0); then
            //     r^2 - dist(centre, line)^2 = (c.e)^2 - K + 2 c.f - |f|^2          (= disc / a of main.odin:249),
            //     h = c.e (3 operations),  w = fma(cz, 2fz, fma(cy, 2fy, fma(cx, 2fx, fma(h, h, -K)))) (4),
            // and the sphere is kept unless w < |f|^2.  Records are stored as PAIRS of spheres
            // {cx0,cx1, cy0,cy1, cz0,cz1, -K0,-K1} and every operation is a packed one (v_pk_fma_f32) on two spheres
            // at once: a plain VALU instruction with an SGPR source issues at half rate on gfx950, a packed one with
            // an SGPR-pair source at full rate; -K is the addend of the square, the one place where an operation has
            // both factors in vector registers and can take the record's pair as its third operand.  Pad records
            // have -K = -FLT_MAX.
            // Error bound: with u = 2^-24, S = |centre| + |o| + r and everything exact but the f32 roundings,
            // w - |f|^2 differs from the exact value by at most 43 u S^2: 18 from rounding d, o, centre and r to f32
            // and from the roundings of e and f (the line moves by at most 8 u S near a sphere it
            // meets), 10 because |e|^2 - 1 and f.e are not exactly zero (9 u and 13 u |o|), 10 from evaluating h and
            // w, 4 from |f|^2, 1 from the difference.  S^2 <= 3(|centre|^2 + r^2 + |o|^2), so lowering K and |f|^2 by
            // 2^-16 (|centre|^2 + r^2) and 2^-16 |o|^2 = 85 u S^2 in all covers it with 42 u S^2 to spare; the margin
            // costs no instruction per sphere: the sphere's share is in its record (which holds -K', rounded up), the
            // ray's share in the threshold.  The spare part makes a second decision certain: hs = h - o.e is the centre's
            // parameter relative to the origin, off by at most 12 u S, so hs < -behind = -2^-19 (reach + |o|_1) is
            // certainly negative, and hs^2 > w - thr then means c = hs^2 - disc/a of main.odin:248 is positive (24 u S^2
            // for the error of hs^2): the line meets the sphere only behind an origin outside it.  The bound needs
            // every product to stay a normal f32 number: the world is checked for that on the host
            // (2^-20 <= r, |centre| + r <= 2^20; the reach passed for any other world is FLT_MAX) and a ray outside
            // 2^-40 <= |d|^2 <= 2^40, |o|^2 <= 2^40 (or not a number) keeps every sphere.  Division and square root are
            // IEEE, so the CPU restatement forms the same values.
            float o_o, fa;
            V3<float> fd, fo;
            V3<float> fe, f2; // unit direction, twice the foot point
            if constexpr (FMA) { fd = rd; fo = ro; fa = a; }
            else {
                fd = {static_cast<float>(rd.x), static_cast<float>(rd.y), static_cast<float>(rd.z)};
                fo = {static_cast<float>(ro.x), static_cast<float>(ro.y), static_cast<float>(ro.z)};
                fa = dot(fd, fd);
            }
            o_o = dot(fo, fo);
            // Throughput mode, the sphere this segment starts on: its origin is the previous hit point, so c is
            // zero up to the rounding of that point.  The reference's f64 arithmetic sees |c| ~ 1e-15
            // and its near root falls below t_min; in f32 the rounding of c is large enough to give
            // spurious hits just above t_min for short or grazing directions.  Take c = 0 for this
            // sphere: the roots of main.odin:249-258 become 0 and 2h/a, and only the second can be
            // a hit, for h > 0 (a segment through the inside).  It is tested first and left out
            // of the scan.
            if constexpr (FMA) {
            if (start_sphere >= 0) {
                const float ocx = start_center.x - ro.x, ocy = start_center.y - ro.y, ocz = start_center.z - ro.z;
                const float q = fma_rn(ocz, rd.z, fma_rn(ocy, rd.y, ocx * rd.x)) * inv_a; // (d.oc)/a as in `resolve`
                const float root = q + q;
                if (q > 0.0f && root >= t_min && root < closest) { closest = root; hit = start_sphere; }
            }
            }
            {
                const float nd = sqrt_rn(fa), inv_nd = 1.0f / nd;
                fe = fd * inv_nd;
                s_o = dot(fo, fe);
                const V3<float> ft = {fma_rn(-s_o, fe.x, fo.x), fma_rn(-s_o, fe.y, fo.y), fma_rn(-s_o, fe.z, fo.z)};
                thr_l = fma_rn(-1.52587890625e-05f, o_o, dot(ft, ft)); // |f|^2 lowered by the ray's share of the margin, 2^-16 |o|^2
                f2 = {ft.x + ft.x, ft.y + ft.y, ft.z + ft.z};
                reach_o = P.filter_reach + (abs_rn(fo.x) + abs_rn(fo.y) + abs_rn(fo.z));
                behind = 1.9073486328125e-06f * reach_o; // 2^-19 = 32 u
                // Mixed mode, the sphere this segment starts on.  Its origin p = o' + t d' was a root of that sphere's
                // equation, so c = |centre - p|^2 - r^2 (main.odin:248) is zero up to rounding: |c| <= 64 * 2^-53 (reach + |o'|)^2
                // (21 from the root and the hit point, 4 from evaluating c; o' the previous origin).  With the
                // centre's parameter certainly below -behind/2 per unit direction, h of main.odin:247 is below
                // -|d| behind/2, and the roots of main.odin:253-257 are <= 0 (c >= 0) or the far one is
                // c/(h - sqrt(disc)) <= |c| / (2|h|) <= |c| / (|d| behind); below t_min the sphere cannot be hit and
                // ray_sphere_intersection need not be evaluated for it.
                if constexpr (MIXED) skip_sphere = (start_sphere >= 0 && start_cc_bound < 9.0e-4f * (nd * behind)) ? start_sphere : -1;
                const bool in_range = (fa >= 9.094947017729282e-13f) && (fa <= 1.099511627776e12f) && (o_o <= 1.099511627776e12f) && // 2^-40, 2^40
                                      (P.filter_reach <= 1048576.0f);
                if (!in_range) {
                    // every sphere is a candidate of this ray: h = 0, w = -K', threshold the lowest number
                    fe = {0.0f, 0.0f, 0.0f}; f2 = {0.0f, 0.0f, 0.0f}; s_o = 0.0f;
                    thr_l = -Lim<float>::max_v;
                    behind = Lim<float>::max_v;
                    skip_sphere = -1;
                    reach_o = Lim<float>::max_v;
                }
            }
            if constexpr (BVH != 0) {
                // Closest hit through the bounding-volume hierarchy instead of the list (the TODO at
                // main.odin:559).  Per lane, no stack: nodes are in depth-first order for this ray's
                // direction octant; a node whose box the segment [0, closest] meets is followed by
                // the next node (its near child), any other by its skip link.  A leaf's spheres go
                // through exactly the two stages of the list scan (same operations, same order), and
                // among equal roots the lowest list index wins as in the list's strict `<`
                // (main.odin:542), so the result is the list scan's bit for bit: the boxes are padded
                // well beyond the rounding of the root evaluation, hence every sphere that
                // evaluation reports as hit is reached.
                // one sphere through the two stages of the list scan (same operations, same order)
                float closest_up = Lim<float>::max_v; // mixed mode: `closest` rounded up to f32, for the box tests
                auto test_sphere = [&](const float cx, const float cy, const float cz, const float K, const float r2, const int index) {
                    ++bvh_tests;
                    // stage 1: the list scan's conservative filter (the record holds -K' in place of K); a NaN keeps the sphere
                    const float h = fma_rn(cz, fe.z, fma_rn(cy, fe.y, cx * fe.x));
                    const float w = fma_rn(cz, f2.z, fma_rn(cy, f2.y, fma_rn(cx, f2.x, fma_rn(h, h, K))));
                    if (w < thr_l) return;
                    const float hs = h - s_o;
                    if ((hs < -behind) && ((fma_rn(hs, hs, -(w - thr_l)) > 0.0f) || (index == skip_sphere))) return;
                    if constexpr (MIXED) {
                        // ray_sphere_intersection as written (main.odin:245-261) in f64 and the caller's strict `<`
                        // (:542-545); equal roots go to the lower list index, which the list reaches first
                        (void)r2;
#ifdef RT_DIAG_COUNTERS
                        if (rank_in(__builtin_amdgcn_ballot_w64(true)) == 0u) ++diag_parked; // one lane per execution by the wave (stats[7]; 5 and 6 hold the hierarchy's counts)
#endif
                        const R4 sp = static_cast<const R4 *>(P.sphere_center)[index]; // {centre, r*r}
                        const Real ocx = sp.x - ro.x, ocy = sp.y - ro.y, ocz = sp.z - ro.z;
                        const Real hh = rd.x * ocx + rd.y * ocy + rd.z * ocz;
                        const Real cc = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w;
                        const Real dd = hh * hh - a * cc;
                        if (dd < Real(0)) return;
                        const Real sq = sqrt_rn(dd);
                        Real root = (hh - sq) / a;
                        if (!(root >= t_min)) root = (hh + sq) / a;
                        if (!(root >= t_min)) return;
                        if (root < closest || (root == closest && index < hit)) {
                            closest = root; hit = index;
                            closest_up = __double2float_ru(root);
                        }
                    } else {
                    if (index == start_sphere) return; // tested before the scan
                    // stage 2: the roots of `resolve`
                    const float ocx = cx - ro.x, ocy = cy - ro.y, ocz = cz - ro.z;
                    const float q = fma_rn(ocz, rd.z, fma_rn(ocy, rd.y, ocx * rd.x)) * inv_a;
                    const float lx = fma_rn(-q, rd.x, ocx), ly = fma_rn(-q, rd.y, ocy), lz = fma_rn(-q, rd.z, ocz);
                    const float D = r2 - fma_rn(lz, lz, fma_rn(ly, ly, lx * lx));
                    if (!(D >= 0.0f)) return;
                    const float sq = sqrt_rn(D * inv_a);
                    float root = q - sq;
                    if (root < t_min) root = q + sq; // the near root is behind t_min: the far one (main.odin:253-257)
                    if (root < t_min) return;
                    // in list order the first sphere of the smallest root wins; the start sphere was tested before the list
                    if (root < closest || (root == closest && hit != start_sphere && index < hit)) { closest = root; hit = index; }
                    }
                };
                // spheres kept outside the hierarchy (far larger than the rest): every segment tests them
                for (int g = 0; g < P.bvh_n_big; ++g)
                    test_sphere(P.bvh_big[g][0], P.bvh_big[g][1], P.bvh_big[g][2], P.bvh_big[g][3], P.bvh_big[g][4], __float_as_int(P.bvh_big[g][6]));

                const int octant =
#ifdef RT_DIAG_BVH_ONE_ORDER
                    0;
#else
                    (fd.x < 0.0f ? 1 : 0) | (fd.y < 0.0f ? 2 : 0) | (fd.z < 0.0f ? 4 : 0);
#endif
                const float4 *__restrict__ leaves = static_cast<const float4 *>(P.bvh_spheres);
                // box corner = origin + q*step, so its slab parameter (corner - o)/d is q*A + B
                const float ix = 1.0f / fd.x, iy = 1.0f / fd.y, iz = 1.0f / fd.z;
                float ax = P.bvh_step[0] * ix, ay = P.bvh_step[1] * iy, az = P.bvh_step[2] * iz;
                float bx = (P.bvh_origin[0] - fo.x) * ix, by = (P.bvh_origin[1] - fo.y) * iy, bz = (P.bvh_origin[2] - fo.z) * iz;
                // The slab parameters below are off by at most 4 u (|q A| + |B|) each (u = 2^-24: the reciprocal, the two
                // products, the difference origin - o and the fused operation), which for an origin far from the scene
                // exceeds what the padding of the boxes absorbs.  The interval test is widened by 2^-20 of the largest
                // |q| |A| + |B| over the axes the ray moves along, so no box whose exact test succeeds is skipped.
                // Mixed mode: the ray itself was rounded from f64, which moves a slab parameter by up to
                // u (|o_axis / d_axis| + |parameter|); the first term joins the sum.  An axis the ray (nearly) does not move
                // along, |1/d_axis| > 2^60 or not a number, is left out of the test there, which only widens it, and a
                // ray outside the filter's range visits every box.
                float ox_i = 0.0f, oy_i = 0.0f, oz_i = 0.0f;
                if constexpr (MIXED) {
                    const float qnan = __int_as_float(0x7fc00000);
                    if (!(abs_rn(ix) <= 1.152921504606847e18f)) { ax = 0.0f; bx = qnan; } else ox_i = abs_rn(fo.x * ix);
                    if (!(abs_rn(iy) <= 1.152921504606847e18f)) { ay = 0.0f; by = qnan; } else oy_i = abs_rn(fo.y * iy);
                    if (!(abs_rn(iz) <= 1.152921504606847e18f)) { az = 0.0f; bz = qnan; } else oz_i = abs_rn(fo.z * iz);
                }
                auto slack_of = [](const float a, const float b, const float c) { const float e = fma_rn(65535.0f, abs_rn(a), abs_rn(b)) + c; return (e < Lim<float>::max_v) ? e : 0.0f; };
                float box_slack = 9.5367431640625e-07f * max_num(max_num(slack_of(ax, bx, ox_i), slack_of(ay, by, oy_i)), slack_of(az, bz, oz_i));
                if (behind == Lim<float>::max_v) box_slack = Lim<float>::max_v; // the ray is outside the filter's range: every box
                // does the segment [0, closest] meet the box {lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16}?
                // A NaN (0 * inf: an axis the ray does not move along) is ignored by min/max, which only widens the box.
                auto meets_box = [&](const uint32_t w0, const uint32_t w1, const uint32_t w2) -> bool {
                    ++bvh_nodes;
#if defined(RT_DIAG_BVH) && !defined(RT_DIAG_BVH_LEAF)
                    if (rank_in(__builtin_amdgcn_ballot_w64(true)) == 0u) ++bvh_wave_steps; // one lane per trip of the wave
#endif
                    const float tx0 = fma_rn(static_cast<float>(w0 & 0xffffu), ax, bx), tx1 = fma_rn(static_cast<float>(w1 >> 16), ax, bx);
                    const float ty0 = fma_rn(static_cast<float>(w0 >> 16), ay, by), ty1 = fma_rn(static_cast<float>(w2 & 0xffffu), ay, by);
                    const float tz0 = fma_rn(static_cast<float>(w1 & 0xffffu), az, bz), tz1 = fma_rn(static_cast<float>(w2 >> 16), az, bz);
                    const float t_in = max_num(max_num(min_num(tx0, tx1), min_num(ty0, ty1)), max_num(min_num(tz0, tz1), 0.0f));
                    float t_max_f;
                    if constexpr (MIXED) t_max_f = closest_up; else t_max_f = closest;
                    const float t_out = min_num(min_num(max_num(tx0, tx1), max_num(ty0, ty1)), min_num(max_num(tz0, tz1), t_max_f));
                    return t_in <= t_out + box_slack;
                };
                auto test_leaf = [&](const int first, const int count) {
#ifdef RT_DIAG_BVH_LEAF
                    if (rank_in(__builtin_amdgcn_ballot_w64(true)) == 0u) ++bvh_wave_steps; // wave-level entries into the leaf code
#endif
                    for (int j = first; j < first + count; ++j) {
                        const float4 s0 = leaves[2 * j], s1 = leaves[2 * j + 1]; // {centre, K} {r*r, 1/r, index, material}
                        test_sphere(s0.x, s0.y, s0.z, s0.w, s1.x, __float_as_int(s1.z));
                    }
                };
                if constexpr (BVH == 1) {
                    const uint4 *__restrict__ nodes = static_cast<const uint4 *>(P.bvh_nodes) + static_cast<size_t>(octant) * static_cast<size_t>(P.bvh_n_nodes);
                    int node = 0;
                    while (node < P.bvh_n_nodes) {
                        const uint4 nd = nodes[node]; // box, link
                        const bool is_leaf = (nd.w & 0x80000000u) != 0u;
                        if (meets_box(nd.x, nd.y, nd.z)) {
                            if (is_leaf) test_leaf(static_cast<int>(nd.w & 0x0fffffffu), static_cast<int>((nd.w >> 28) & 7u) + 1);
                            node = node + 1;
                        } else {
                            node = is_leaf ? node + 1 : static_cast<int>(nd.w);
                        }
                    }
                } else {
                    // LDS form: tree order, the successors of a node for this octant in one 32-bit word
                    const uint32_t *links = l_links + octant * P.bvh_n_nodes;
                    uint32_t node = P.bvh_n_nodes > 0 ? 0u : 0xffffu;
                    while (node != 0xffffu) {
                        const uint4 nd = l_boxes[node]; // box, leaf range
                        const uint32_t link = links[node];
                        if (meets_box(nd.x, nd.y, nd.z)) {
                            if (nd.w != 0u) test_leaf(static_cast<int>(nd.w & 0x1fffu), static_cast<int>((nd.w >> 13) & 3u) + 1);
                            node = link & 0xffffu;
                        } else {
                            node = link >> 16;
                        }
                    }
                }
            } else {
            // per-ray factors, both halves alike
            const F2 ex2 = {fe.x, fe.x}, ey2 = {fe.y, fe.y}, ez2 = {fe.z, fe.z};
            const F2 fx2 = {f2.x, f2.x}, fy2 = {f2.y, f2.y}, fz2 = {f2.z, f2.z};
            auto test_pairs = [&](const Pair &p, const Pair &q, const int i) {
                // h = c.e, w = (h*h - K') + 2 c.f: seven packed operations per pair of spheres (p.k holds -K')
                const F2 hp = fma2(p.cz, ez2, fma2(p.cy, ey2, p.cx * ex2));
                const F2 hq = fma2(q.cz, ez2, fma2(q.cy, ey2, q.cx * ex2));
                const F2 dp = fma2(p.cz, fz2, fma2(p.cy, fy2, fma2(p.cx, fx2, fma2(hp, hp, p.k))));
                const F2 dq = fma2(q.cz, fz2, fma2(q.cy, fy2, fma2(q.cx, fx2, fma2(hq, hq, q.k))));
                const float dmax = max_num(max_num(dp.x, dp.y), max_num(dq.x, dq.y));
#ifdef RT_DIAG_NO_PARK
                asm volatile("" :: "v"(dmax));
                if (false) {
#else
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(dmax >= thr_l) != 0ull, 0)) {
#endif
                    RT_DIAG_COUNT(diag_entries);
#ifdef RT_DIAG_PHASES
                    ph_in = __builtin_amdgcn_s_memtime();
#endif
                    park(hp.x, 0.0f, dp.x, i); park(hp.y, 0.0f, dp.y, i + 1); park(hq.x, 0.0f, dq.x, i + 2); park(hq.y, 0.0f, dq.y, i + 3);
#ifdef RT_DIAG_PHASES
                    ph_slow += __builtin_amdgcn_s_memtime() - ph_in;
#endif
                }
            };
            // two register sets, A and B, each holding one group: while one is tested the other loads
            Pair a0 = pair(0), a1 = pair(1);
            for (int i = 0; i < n; i += 2 * GROUP) {
                const Pair b0 = pair(i / 2 + 2), b1 = pair(i / 2 + 3);
                test_pairs(a0, a1, i);
                a0 = pair(i / 2 + 4); a1 = pair(i / 2 + 5);
                test_pairs(b0, b1, i + 4);
            }
            } // list scan
        } else {
            // literal modes: records are {centre, r*r}, the reference's expressions (main.odin:245-249)
            auto discriminant = [&](const R4 sp, Real &h, Real &cc) -> Real {
                const Real ocx = sp.x - ro.x, ocy = sp.y - ro.y, ocz = sp.z - ro.z;
                h = rd.x * ocx + rd.y * ocy + rd.z * ocz;
                cc = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w;
                return h * h - a * cc;
            };
            auto test_group = [&](const R4 &s0, const R4 &s1, const R4 &s2, const R4 &s3, const int i) {
                Real h0, h1, h2, h3, k0, k1, k2, k3;
                const Real d0 = discriminant(s0, h0, k0), d1 = discriminant(s1, h1, k1);
                const Real d2 = discriminant(s2, h2, k2), d3 = discriminant(s3, h3, k3);
                const Real dmax = max_num(max_num(d0, d1), max_num(d2, d3));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(dmax >= Real(0)) != 0ull, 0)) {
                    park(h0, k0, d0, i); park(h1, k1, d1, i + 1); park(h2, k2, d2, i + 2); park(h3, k3, d3, i + 3);
                }
            };
            R4 a0 = rec(0), a1 = rec(1), a2 = rec(2), a3 = rec(3);
            for (int i = 0; i < n; i += 2 * GROUP) {
                const R4 b0 = rec(i + 4), b1 = rec(i + 5), b2 = rec(i + 6), b3 = rec(i + 7);
                test_group(a0, a1, a2, a3, i);
                a0 = rec(i + 8); a1 = rec(i + 9); a2 = rec(i + 10); a3 = rec(i + 11);
                test_group(b0, b1, b2, b3, i + 4);
            }
        }
#ifdef RT_DIAG_PHASES
        ph_in = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (WQ) { if (qcount != 0u) flush(); }
        else if (pend >= 0) resolve();
#ifdef RT_DIAG_PHASES
        ph_slow += __builtin_amdgcn_s_memtime() - ph_in;
#endif
        ++segments;
        RT_PHASE(ph_scan);

        // ---------------- shading
        bool ended;
        const bool is_hit = hit >= 0;
        R4 sp = {}, m = {};
        int mtype = -1;
        V3<Real> center = {Real(0), Real(0), Real(0)}, p = center, nrm = center;
        bool front = false;
        if (is_hit) {
            sp = static_cast<const R4 *>(P.sphere_center)[hit];
            const Real radius = static_cast<const Real *>(P.sphere_radius)[hit];
            const int mi = P.sphere_material[hit];
            mtype = P.material_type[mi];
            m = static_cast<const R4 *>(P.material_rec)[mi];
            // hit record (main.odin:553-557)
            center = {sp.x, sp.y, sp.z};
            p = ro + closest * rd;
            nrm = FMA ? (p - center) * radius : (p - center) / radius; // throughput mode: the array holds 1/r
            front = dot(rd, nrm) < Real(0);
            nrm = front ? nrm : -nrm;
        }
        // One normalisation for the three places that normalise once per segment, each lane handing in the vector its
        // own path normalises (so every lane computes exactly what it would compute in its own branch, in one pass
        // of the wave through the square root and the divisions instead of three): the background's
        // normalize(ray.direction) (main.odin:570), the dielectric's (:373) and the metals' normalize(reflect(...)) (:320).
        // (list scan only: through the hierarchy, where the kernels are built for more waves per SIMD, the longer live
        // ranges cost more in spilled registers than the shared pass saves; each branch normalises for itself there)
        constexpr bool SHARED_NORMALIZE = BVH == 0;
        V3<Real> unit = rd;
        if constexpr (SHARED_NORMALIZE) {
            if (!is_hit || mtype != 0) unit = normalize<Real, FMA>((is_hit && mtype != 2) ? reflect(rd, nrm) : rd);
        }
        auto normalized = [&](const V3<Real> x) -> V3<Real> { if constexpr (SHARED_NORMALIZE) { (void)x; return unit; } else return normalize<Real, FMA>(x); };
        // random_unit_vector for the lanes that hit a lambertian or a metal (main.odin:299, :321), by the wave together
        uint32_t cand_x = 0u, cand_y = 0u, cand_z = 0u;
        if constexpr (COOP) {
            const bool need = is_hit && mtype != 2;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                const uint32_t kb_scatter = mix32(ks ^ (0x85ebca6bu * (static_cast<uint32_t>(bounce) + 2u))); // the stream of (pixel, sample, bounce + 1)
                first_unit_candidate<Real>(need, kb_scatter, wp, WQ ? wq_entry[wave_in_block] : coop_slot[wave_in_block], lane, cand_x, cand_y, cand_z);
            }
        }
        if (is_hit) {
            Draws g(ks, wp, static_cast<uint32_t>(bounce) + 1u);
            V3<Real> dir;
            V3<Real> atten = {m.x, m.y, m.z};
            bool ok = true;
            if (mtype == 2) {
                // dielectric_proc (main.odin:356-400)
                const Real ri = m.w;
                Real src = (ri >= Real(1)) ? Real(1) : Real(1) / ri;
                Real dst = (ri >= Real(1)) ? ri : Real(1);
                if (!front) { Real t = src; src = dst; dst = t; }
                const Real rel = src / dst;
                const V3<Real> unit_d = normalized(rd);
                Real cos_theta = dot(-unit_d, nrm);
                cos_theta = min_std(cos_theta, Real(1));
                const Real sin2 = Real(1) - cos_theta * cos_theta;
                const bool must_reflect = (rel * rel * sin2) > Real(1);
                bool do_reflect = must_reflect;
                if (!must_reflect) {
                    // Schlick (main.odin:111-122); the uniform is drawn only on this branch (:390)
                    Real r0 = (Real(1) - rel) / (Real(1) + rel);
                    r0 *= r0;
                    const Real qq = Real(1) - cos_theta;
                    const Real refl = r0 + (Real(1) - r0) * qq * qq * qq * qq * qq;
                    do_reflect = refl > g.next<Real>();
                }
                dir = do_reflect ? reflect(unit_d, nrm) : refract(unit_d, nrm, rel);
                atten = {Real(1), Real(1), Real(1)};
            } else {
                V3<Real> ruv;
                if constexpr (COOP) ruv = unit_vector_of_candidate<Real, FMA>(cand_x, cand_y, cand_z);
                else ruv = random_unit_vector<Real, FMA>(g);
                if (mtype == 0) {
                    // lambertian_proc (main.odin:292-312)
                    dir = nrm + ruv;
                    const Real EPS = static_cast<Real>(1e-8);
                    if (abs_rn(dir.x) <= EPS && abs_rn(dir.y) <= EPS && abs_rn(dir.z) <= EPS) dir = nrm;
                } else {
                    // metallic_proc without the Fresnel block (main.odin:315-323, :348-353)
                    dir = normalized(reflect(rd, nrm));
                    dir = dir + m.w * ruv;
                    if (mtype >= 4) {
                        // USE_METAL_FRESNEL with METAL_FRESNEL_KIND 0 (type 4: Schlick, main.odin:334-335, :111-122) or 1
                        // (Hoffman, :337-339, :132-144).  Type 5 is kind 1 as the reference's text computes it: the Schlick
                        // term with its arguments swapped (:133 passes (r0, cos_i)), cos + (1 - cos)(1 - r0)^5, and the
                        // factors 823543 / 46656 and 49 / 6 as the integer quotients 17 and 8 (:142).  Type 6 is the
                        // approximation of the cited papers: arguments in declared order, 7^7/6^6 and 49/6 as real numbers.
                        // 1.0/1.27035 is a constant expression there, i.e. the exact quotient rounded once; math.pow(x, 6)
                        // is taken as ((x*x)*x)^2.
                        Real cos_theta = dot(-normalize<Real, FMA>(rd), nrm);
                        cos_theta = min_std(cos_theta, Real(1));
                        const Real rel = static_cast<Real>(0.7871846341559413);
                        Real r0 = (Real(1) - rel) / (Real(1) + rel);
                        r0 *= r0;
                        const Real qq = Real(1) - cos_theta;
                        Real refl = r0 + (Real(1) - r0) * qq * qq * qq * qq * qq;
                        if (mtype >= 5) {
                            const bool paper = mtype == 6;
                            const Real aa = (paper ? static_cast<Real>(823543.0 / 46656.0) : Real(17)) * (r0 - Real(0.5)) +
                                            (paper ? static_cast<Real>(49.0 / 6.0) : Real(8)) * (Real(1) - r0);
                            if (!paper) {
                                const Real a1 = Real(1) - r0; // reflectance_schlick_approximation(cos_i := r0, r0 := cos_theta)
                                refl = cos_theta + (Real(1) - cos_theta) * a1 * a1 * a1 * a1 * a1;
                            }
                            const Real q2 = qq * qq, q3 = q2 * qq;
                            refl = refl - aa * cos_theta * (q3 * q3);
                        }
                        atten = atten * (Real(1) - refl) + V3<Real>{Real(1), Real(1), Real(1)} * refl;
                    } else if (mtype == 3) {
                        // USE_METAL_FRESNEL, METAL_FRESNEL_KIND 2 (main.odin:326-328, :341-347, :84-109)
                        Real cos_theta = dot(-normalize<Real, FMA>(rd), nrm);
                        cos_theta = min_std(cos_theta, Real(1));
                        const Real sin_theta = sqrt_rn(Real(1) - cos_theta * cos_theta);
                        const Real n1 = Real(1), n2 = static_cast<Real>(1.27035);
                        Real sub = (n1 / n2) * sin_theta;
                        sub = Real(1) - sub * sub;
                        sub = sqrt_rn(sub);
                        Real x1 = n1 * cos_theta, x2 = n2 * sub;
                        Real rs = (x1 - x2);
                        rs = rs / (x1 + x2);
                        rs = rs * rs;
                        x1 = n1 * sub; x2 = n2 * cos_theta;
                        Real rp = (x1 - x2);
                        rp = rp / (x1 + x2);
                        rp = rp * rp;
                        const Real refl = (rs + rp) * Real(0.5);
                        // math.lerp(albedo, {1,1,1}, t) = albedo*(1-t) + 1*t
                        atten = atten * (Real(1) - refl) + V3<Real>{Real(1), Real(1), Real(1)} * refl;
                    }
                    ok = dot(dir, nrm) > Real(0);
                }
            }
            if (ok) {
                ro = p; rd = dir;
                start_sphere = hit; start_center = center;
                if constexpr (MIXED) start_cc_bound = 7.2e-15f * reach_o * reach_o; // 64 * 2^-53 = 7.1e-15, rounded up
                color = color * atten;
                ++bounce;
                ended = bounce > P.max_bounces; // `for _ in 0..=max_ray_bounces` ran out: colour is kept (main.odin:535)
            } else {
                color = {Real(0), Real(0), Real(0)};
                ended = true;
            }
            RT_PHASE(ph_hit);
        } else {
            // background (main.odin:570-572, :412-418): lerp(a, b, t) = a*(1-t) + b*t
            const V3<Real> d = normalized(rd);
            const Real t = Real(0.5) * (d.y + Real(1));
            const V3<Real> wa = {Real(1), Real(1), Real(1)};
            const V3<Real> wb = {static_cast<Real>(0.5), static_cast<Real>(0.7), static_cast<Real>(1.0)};
            color = color * (wa * (Real(1) - t) + wb * t);
            ended = true;
        }
        if (ended) {
            // pixel_color += sample_color (main.odin:575)
            if constexpr (LSUM) {
                // fixed-point colour -> the tile's sums in LDS (or HBM directly when the tile's set has gone out already)
                unsigned long long w[SUM_WORDS];
                if constexpr (MIXED && FIXED) { to_q32_64(color.x, w[0], w[1]); to_q32_64(color.y, w[2], w[3]); to_q32_64(color.z, w[4], w[5]); }
                else { w[0] = to_q32_32(color.x); w[1] = to_q32_32(color.y); w[2] = to_q32_32(color.z); }
                const uint32_t age = cur_gen - (my_slot >> 6);
                if (age < static_cast<uint32_t>(GENS)) {
                    unsigned long long *dst = &tile_sum[wave_in_block][GENS == 2 ? ((my_slot >> 6) & 1u) : 0u][(my_slot & 63u) * SUM_WORDS];
#pragma unroll
                    for (int c = 0; c < SUM_WORDS; ++c) atomicAdd(dst + c, w[c]);
                } else {
                    unsigned long long *dst = static_cast<unsigned long long *>(P.accum) + static_cast<int64_t>(pixel) * SUM_WORDS;
#pragma unroll
                    for (int c = 0; c < SUM_WORDS; ++c) atomicAdd(dst + c, w[c]);
                }
            } else if constexpr (MIXED) {
                // the sample's colour goes to its own slot; order_sum_kernel adds the slots in sample order
                double *dst = static_cast<double *>(P.slabs) + (static_cast<int64_t>(s) * P.slab_stride + stage_px);
                dst[0] = color.x; dst[1] = color.y; dst[2] = color.z;
            } else if constexpr (QUEUE) {
                unsigned long long *dst = static_cast<unsigned long long *>(P.accum) + static_cast<int64_t>(pixel) * 3;
                atomicAdd(dst + 0, to_q32_32(color.x));
                atomicAdd(dst + 1, to_q32_32(color.y));
                atomicAdd(dst + 2, to_q32_32(color.z));
            } else {
                acc = acc + color;
            }
            ++s;
            if constexpr (QUEUE) need_unit = true; // the lane asks for another unit
            else fresh = true;
        }
        RT_PHASE(ph_end);
    }

    if constexpr (!QUEUE) {
        if (inside) {
            const uint32_t chunk = blockIdx.y;
            Real *out = (chunk == 0 ? static_cast<Real *>(P.accum) : static_cast<Real *>(P.slabs) + static_cast<int64_t>(chunk - 1) * P.slab_stride) +
                        static_cast<int64_t>(pixel) * 3;
            out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
        }
    }
    if (P.stats != nullptr) {
        // wave totals: segments traced, camera rays started, and the wave's loop trip count
        // (a lane that is alive in every iteration exists in both modes: the last to finish)
        uint32_t seg = segments, iters = segments, smp = started;
        for (int off = 32; off > 0; off >>= 1) {
            seg += __shfl_xor(seg, off, 64);
            smp += __shfl_xor(smp, off, 64);
            iters = max(iters, static_cast<uint32_t>(__shfl_xor(iters, off, 64)));
        }
        if (lane == 0) {
            atomicAdd(&P.stats[0], static_cast<unsigned long long>(seg));
            atomicAdd(&P.stats[3], static_cast<unsigned long long>(smp));
            atomicAdd(&P.stats[4], static_cast<unsigned long long>(iters));
        }
        if constexpr (BVH != 0) {
            uint32_t nb = bvh_nodes, nt = bvh_tests;
            for (int off = 32; off > 0; off >>= 1) { nb += __shfl_xor(nb, off, 64); nt += __shfl_xor(nt, off, 64); }
            if (lane == 0) {
                atomicAdd(&P.stats[5], static_cast<unsigned long long>(nb));
                atomicAdd(&P.stats[6], static_cast<unsigned long long>(nt));
            }
#ifdef RT_DIAG_BVH
            uint32_t ws = bvh_wave_steps;
            for (int off = 32; off > 0; off >>= 1) ws += __shfl_xor(ws, off, 64);
            if (lane == 0) atomicAdd(&P.stats[7], static_cast<unsigned long long>(ws));
#endif
        }
#ifdef RT_DIAG_PHASES
        if (lane == 0) {
            atomicAdd(&P.stats[5], ph_fresh);
            atomicAdd(&P.stats[6], ph_scan - ph_slow);
            atomicAdd(&P.stats[7], ph_slow);
#ifdef RT_DIAG_FLUSH
            atomicAdd(&P.stats[1], ph_flush);
#else
            atomicAdd(&P.stats[1], ph_hit);
#endif
            atomicAdd(&P.stats[2], ph_end);
        }
#endif
#ifdef RT_DIAG_COUNTERS
        // entries: per wave (uniform, lane 0 holds it); resolves: executions by the wave (counted by one lane each); parked: lane total
        uint32_t par = diag_parked, res = diag_resolves;
        for (int off = 32; off > 0; off >>= 1) { par += __shfl_xor(par, off, 64); res += __shfl_xor(res, off, 64); }
        if (lane == 0) {
            atomicAdd(&P.stats[5], static_cast<unsigned long long>(diag_entries));
            atomicAdd(&P.stats[6], static_cast<unsigned long long>(res));
            atomicAdd(&P.stats[7], static_cast<unsigned long long>(par));
        }
#endif
    }
}

// ----------------------------------------------------------------------------- small kernels
// thread -> owned pixel: the tile enumeration of render_kernel, 64 pixels per tile
__device__ __forceinline__ bool owned_pixel(const TileMap &m, int min_x, int min_y, int max_x, int max_y, int width, int64_t &index) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int tile = static_cast<int>(gid >> 6);
    const int lane = static_cast<int>(gid & 63);
    if (tile >= m.n_tiles) return false;
    const int u = min_x + (tile % m.tiles_x) * TILE + (lane & 7);
    const int v = min_y + tile_row_of(m, tile / m.tiles_x) * TILE + (lane >> 3);
    if (u >= max_x || v >= max_y) return false;
    index = static_cast<int64_t>(u) + static_cast<int64_t>(v) * width;
    return true;
}

// zero the Q32.32 accumulators of the owned pixels and the queue head, ahead of a queue-mode launch
__global__ void clear_kernel(const ReduceParams P, uint32_t *queue) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *queue = 0u;
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    unsigned long long *a = static_cast<unsigned long long *>(P.accum) + idx * P.words_per_pixel;
    for (int k = 0; k < P.words_per_pixel; ++k) a[k] = 0ull;
}

__global__ void clear_queue_kernel(uint32_t *queue) { *queue = 0u; }

// literal modes: accum[p] += slab_0[p] + slab_1[p] + ... in chunk order
template <typename Real> __global__ void reduce_kernel(const ReduceParams P) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    Real *base = static_cast<Real *>(P.accum) + idx * 3;
    Real x = base[0], y = base[1], z = base[2];
    for (int c = 1; c < P.chunks; ++c) {
        const Real *sl = static_cast<const Real *>(P.slabs) + static_cast<int64_t>(c - 1) * P.slab_stride + idx * 3;
        x = x + sl[0]; y = y + sl[1]; z = z + sl[2];
    }
    base[0] = x; base[1] = y; base[2] = z;
}

// mixed mode: pixel_color += sample_color in sample order (main.odin:575) over the stored colours of
// one launch.  Thread gid owns pixel (gid & 63) of owned tile (gid >> 6), the enumeration the render
// kernel stored under: colour of sample s at staged[(s * stride) + gid * 3].  `chunks` is the number
// of samples staged; first: the sum starts from zero, otherwise from the accumulator's value.
__global__ void order_sum_kernel(const ReduceParams P, const int first) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    double *base = static_cast<double *>(P.accum) + idx * 3;
    double x = 0.0, y = 0.0, z = 0.0;
    if (!first) { x = base[0]; y = base[1]; z = base[2]; }
    const double *sl = static_cast<const double *>(P.slabs) + gid * 3;
    for (int c = 0; c < P.chunks; ++c, sl += P.slab_stride) { x = x + sl[0]; y = y + sl[1]; z = z + sl[2]; }
    base[0] = x; base[1] = y; base[2] = z;
}

// main.odin:577-586: pixel_color *= 1/spp, per-channel sqrt, store
template <typename Real> __global__ void finalize_kernel(const FinalizeParams P) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    const Real *in = static_cast<const Real *>(P.accum) + idx * 3;
    const Real contribution = Real(1) / static_cast<Real>(P.total_samples);
    const Real mx = in[0] * contribution, my = in[1] * contribution, mz = in[2] * contribution;
    if (P.linear != nullptr) {
        P.linear[idx * 3 + 0] = static_cast<double>(mx);
        P.linear[idx * 3 + 1] = static_cast<double>(my);
        P.linear[idx * 3 + 2] = static_cast<double>(mz);
    }
    P.image[idx * 3 + 0] = static_cast<double>(sqrt_rn(mx));
    P.image[idx * 3 + 1] = static_cast<double>(sqrt_rn(my));
    P.image[idx * 3 + 2] = static_cast<double>(sqrt_rn(mz));
}

// The same tail for Q32.32 sums, in f64: u64 -> f64 with one rounding, * 2^-32, * 1/spp, sqrt.
__device__ __forceinline__ double q32_32_to_f64(unsigned long long v) {
    return static_cast<double>(static_cast<uint32_t>(v >> 32)) * 4294967296.0 + static_cast<double>(static_cast<uint32_t>(v));
}
__global__ void finalize_q_kernel(const FinalizeParams P) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    const unsigned long long *in = static_cast<const unsigned long long *>(P.accum) + idx * 3;
    const double contribution = 1.0 / static_cast<double>(P.total_samples);
    for (int c = 0; c < 3; ++c) {
        const double mean = (q32_32_to_f64(in[c]) * (1.0 / 4294967296.0)) * contribution;
        if (P.linear != nullptr) P.linear[idx * 3 + c] = mean;
        P.image[idx * 3 + c] = sqrt_rn(mean);
    }
}

// The same tail for Q32.64 sums: the sum rounded to f64 once, * 1/spp, sqrt.
__global__ void finalize_q2_kernel(const FinalizeParams P) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    const unsigned long long *in = static_cast<const unsigned long long *>(P.accum) + idx * 6;
    const double contribution = 1.0 / static_cast<double>(P.total_samples);
    for (int c = 0; c < 3; ++c) {
        const double mean = q32_64_to_f64(in[2 * c], in[2 * c + 1]) * contribution;
        if (P.linear != nullptr) P.linear[idx * 3 + c] = mean;
        P.image[idx * 3 + c] = sqrt_rn(mean);
    }
}

// serialize_ppm's body on the device (main.odin:806-820): each channel is
// u8(256 * min(x, 0.999)) printed right-aligned in three columns and followed by ' ' (after blue:
// '\n'), i.e. one little-endian 32-bit word per channel.  The gamma value x is formed exactly as
// finalize_kernel / finalize_q_kernel form it, so the text equals that of the host writer run on
// the f64 image.
__device__ __forceinline__ uint32_t ppm_word(double x, bool last) {
    const double clipped = (0.999 < x) ? 0.999 : x;
    const uint32_t v = static_cast<uint32_t>(static_cast<uint8_t>(static_cast<long long>(256.0 * clipped)));
    const uint32_t hundreds = v / 100u, tens = (v / 10u) % 10u, ones = v % 10u;
    const uint32_t c0 = v >= 100u ? 0x30u + hundreds : 0x20u;
    const uint32_t c1 = v >= 10u ? 0x30u + tens : 0x20u;
    return c0 | (c1 << 8) | ((0x30u + ones) << 16) | ((last ? 0x0au : 0x20u) << 24);
}
template <int ARITH> __global__ void finalize_ppm_kernel(const PpmParams P) {
    int64_t idx;
    if (!owned_pixel(P.map, P.min_x, P.min_y, P.max_x, P.max_y, P.width, idx)) return;
    double g[3];
    if constexpr (ARITH == ARITH_F32_FMA) {
        const unsigned long long *in = static_cast<const unsigned long long *>(P.accum) + idx * 3;
        const double contribution = 1.0 / static_cast<double>(P.total_samples);
        for (int c = 0; c < 3; ++c) g[c] = sqrt_rn((q32_32_to_f64(in[c]) * (1.0 / 4294967296.0)) * contribution);
    } else if constexpr (ARITH == ARITH_MIXED_FIXED) {
        const unsigned long long *in = static_cast<const unsigned long long *>(P.accum) + idx * 6;
        const double contribution = 1.0 / static_cast<double>(P.total_samples);
        for (int c = 0; c < 3; ++c) g[c] = sqrt_rn(q32_64_to_f64(in[2 * c], in[2 * c + 1]) * contribution);
    } else {
        using Real = typename std::conditional<ARITH == ARITH_F64, double, float>::type;
        const Real *in = static_cast<const Real *>(P.accum) + idx * 3;
        const Real contribution = Real(1) / static_cast<Real>(P.total_samples);
        for (int c = 0; c < 3; ++c) g[c] = static_cast<double>(sqrt_rn(in[c] * contribution));
    }
    uint32_t *out = P.text + idx * 3;
    out[0] = ppm_word(g[0], false); out[1] = ppm_word(g[1], false); out[2] = ppm_word(g[2], true);
}

// ----------------------------------------------------------------------------- static scratch
// The array's name is a host-side handle, not a device address: each GPU holds its own copy at its
// own address, looked up once per device and cached; kernels and memsets only ever get that pointer.
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

hipError_t static_scratch_address(int device, void **out) {
    static std::mutex mu;
    static void *address[64] = {};
    if (out == nullptr || device < 0 || device >= 64) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(mu);
    if (address[device] == nullptr) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice; // the symbol resolves on the current device
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        address[device] = ptr;
    }
    *out = address[device];
    return hipSuccess;
}

// ----------------------------------------------------------------------------- launchers
template <class K> static void launch_k(K kernel, dim3 grid, int fetch, int n_records, size_t rec_bytes, hipStream_t stream, const RenderParams &p) {
    const size_t lds = fetch == FETCH_LDS ? static_cast<size_t>(n_records + GROUP) * rec_bytes : 0;
    hipLaunchKernelGGL(kernel, grid, dim3(BLOCK_THREADS, 1, 1), lds, stream, p);
}

hipError_t launch_render(const RenderParams &p, int arith, int fetch, int chunks, int resident_blocks, hipStream_t stream) {
    if (p.map.n_tiles <= 0 || chunks <= 0) return hipSuccess;
    const int blocks = (p.map.n_tiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    const bool lds = fetch == FETCH_LDS;
    switch (arith) {
    case ARITH_F32_FMA: {
        // queue mode: as many workgroups as the chip holds at once (or fewer if the work is smaller)
        const int64_t waves_needed = static_cast<int64_t>(p.n_items);
        const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(resident_blocks, (waves_needed + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)));
        if (fetch == FETCH_BVH && p.bvh_lds_boxes != nullptr) {
            // the hierarchy fits LDS: 48 bytes per node and workgroup, which also bounds how many workgroups a CU holds
            const size_t lds_bytes = static_cast<size_t>(p.bvh_n_nodes) * 48;
            const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(2048 / BVH_LDS_BLOCK_THREADS, (160 * 1024) / std::max<size_t>(lds_bytes, 1))));
            const int fit = std::max(1, resident_blocks / 8 * per_cu);
            const int wanted = static_cast<int>((waves_needed + BVH_LDS_BLOCK_THREADS / 64 - 1) / (BVH_LDS_BLOCK_THREADS / 64));
            hipLaunchKernelGGL((render_kernel<float, true, FETCH_SCALAR, true, 2>), dim3(std::max(1, std::min(wanted, fit))), dim3(BVH_LDS_BLOCK_THREADS, 1, 1), lds_bytes, stream, p);
        } else if (fetch == FETCH_BVH) launch_k(render_kernel<float, true, FETCH_SCALAR, true, 1>, dim3(grid), FETCH_SCALAR, p.n_spheres, sizeof(float4), stream, p);
        else if (lds) launch_k(render_kernel<float, true, FETCH_LDS, true>, dim3(grid), fetch, p.n_spheres, sizeof(float4), stream, p);
        else launch_k(render_kernel<float, true, FETCH_SCALAR, true>, dim3(grid), fetch, p.n_spheres, sizeof(float4), stream, p);
        break;
    }
    case ARITH_F64:
        if (lds) launch_k(render_kernel<double, false, FETCH_LDS, false>, dim3(blocks, chunks), fetch, p.n_spheres, sizeof(double4), stream, p);
        else launch_k(render_kernel<double, false, FETCH_SCALAR, false>, dim3(blocks, chunks), fetch, p.n_spheres, sizeof(double4), stream, p);
        break;
    case ARITH_MIXED: {
        const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(resident_blocks, (static_cast<int64_t>(p.n_items) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)));
        if (fetch == FETCH_BVH && p.bvh_lds_boxes != nullptr) {
            // as in the throughput mode; this instantiation is built for six waves per SIMD: three workgroups of 512 per CU
            const size_t lds_bytes = static_cast<size_t>(p.bvh_n_nodes) * 48;
            const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(6 * 256 / BVH_LDS_BLOCK_THREADS, (160 * 1024) / std::max<size_t>(lds_bytes, 1))));
            const int fit = std::max(1, resident_blocks / 8 * per_cu);
            const int wanted = static_cast<int>((static_cast<int64_t>(p.n_items) + BVH_LDS_BLOCK_THREADS / 64 - 1) / (BVH_LDS_BLOCK_THREADS / 64));
            hipLaunchKernelGGL((render_kernel<double, false, FETCH_SCALAR, true, 2, true>), dim3(std::max(1, std::min(wanted, fit))), dim3(BVH_LDS_BLOCK_THREADS, 1, 1), lds_bytes, stream, p);
        } else if (fetch == FETCH_BVH) launch_k(render_kernel<double, false, FETCH_SCALAR, true, 1, true>, dim3(grid), FETCH_SCALAR, p.n_spheres, sizeof(float4), stream, p);
        else launch_k(render_kernel<double, false, FETCH_SCALAR, true, 0, true>, dim3(grid), FETCH_SCALAR, p.n_spheres, sizeof(float4), stream, p);
        break;
    }
    case ARITH_MIXED_FIXED: {
        const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(resident_blocks, (static_cast<int64_t>(p.n_items) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)));
        if (fetch == FETCH_BVH && p.bvh_lds_boxes != nullptr) {
            // this instantiation is built for four waves per SIMD: two workgroups of 512 per CU
            const size_t lds_bytes = static_cast<size_t>(p.bvh_n_nodes) * 48;
            const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(4 * 256 / BVH_LDS_BLOCK_THREADS, (160 * 1024) / std::max<size_t>(lds_bytes, 1))));
            const int fit = std::max(1, resident_blocks / 8 * per_cu);
            const int wanted = static_cast<int>((static_cast<int64_t>(p.n_items) + BVH_LDS_BLOCK_THREADS / 64 - 1) / (BVH_LDS_BLOCK_THREADS / 64));
            hipLaunchKernelGGL((render_kernel<double, false, FETCH_SCALAR, true, 2, true, true>), dim3(std::max(1, std::min(wanted, fit))), dim3(BVH_LDS_BLOCK_THREADS, 1, 1), lds_bytes, stream, p);
        } else if (fetch == FETCH_BVH) launch_k(render_kernel<double, false, FETCH_SCALAR, true, 1, true, true>, dim3(grid), FETCH_SCALAR, p.n_spheres, sizeof(float4), stream, p);
        else launch_k(render_kernel<double, false, FETCH_SCALAR, true, 0, true, true>, dim3(grid), FETCH_SCALAR, p.n_spheres, sizeof(float4), stream, p);
        break;
    }
    case ARITH_F32_LITERAL:
        if (lds) launch_k(render_kernel<float, false, FETCH_LDS, false>, dim3(blocks, chunks), fetch, p.n_spheres, sizeof(float4), stream, p);
        else launch_k(render_kernel<float, false, FETCH_SCALAR, false>, dim3(blocks, chunks), fetch, p.n_spheres, sizeof(float4), stream, p);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static dim3 pixel_grid(const TileMap &m) { return dim3(static_cast<unsigned>((static_cast<int64_t>(m.n_tiles) * 64 + 255) / 256)); }

hipError_t launch_clear(const ReduceParams &p, uint32_t *queue, hipStream_t stream) {
    if (p.map.n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(clear_kernel, pixel_grid(p.map), dim3(256), 0, stream, p, queue);
    return hipGetLastError();
}

hipError_t launch_clear_queue(uint32_t *queue, hipStream_t stream) {
    hipLaunchKernelGGL(clear_queue_kernel, dim3(1), dim3(1), 0, stream, queue);
    return hipGetLastError();
}

hipError_t launch_order_sum(const ReduceParams &p, bool first, hipStream_t stream) {
    if (p.map.n_tiles <= 0 || (p.chunks <= 0 && !first)) return hipSuccess; // no samples at all: the first sum still writes zeros
    hipLaunchKernelGGL(order_sum_kernel, pixel_grid(p.map), dim3(256), 0, stream, p, first ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_reduce(const ReduceParams &p, int arith, hipStream_t stream) {
    if (p.map.n_tiles <= 0 || p.chunks <= 1 || arith == ARITH_F32_FMA || arith == ARITH_MIXED || arith == ARITH_MIXED_FIXED) return hipSuccess;
    if (arith == ARITH_F64) hipLaunchKernelGGL(reduce_kernel<double>, pixel_grid(p.map), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(reduce_kernel<float>, pixel_grid(p.map), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeParams &p, int arith, hipStream_t stream) {
    if (p.map.n_tiles <= 0) return hipSuccess;
    if (arith == ARITH_F64 || arith == ARITH_MIXED) hipLaunchKernelGGL(finalize_kernel<double>, pixel_grid(p.map), dim3(256), 0, stream, p);
    else if (arith == ARITH_MIXED_FIXED) hipLaunchKernelGGL(finalize_q2_kernel, pixel_grid(p.map), dim3(256), 0, stream, p);
    else if (arith == ARITH_F32_FMA) hipLaunchKernelGGL(finalize_q_kernel, pixel_grid(p.map), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(finalize_kernel<float>, pixel_grid(p.map), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_finalize_ppm(const PpmParams &p, int arith, hipStream_t stream) {
    if (p.map.n_tiles <= 0) return hipSuccess;
    if (arith == ARITH_F64 || arith == ARITH_MIXED) hipLaunchKernelGGL(finalize_ppm_kernel<ARITH_F64>, pixel_grid(p.map), dim3(256), 0, stream, p);
    else if (arith == ARITH_MIXED_FIXED) hipLaunchKernelGGL(finalize_ppm_kernel<ARITH_MIXED_FIXED>, pixel_grid(p.map), dim3(256), 0, stream, p);
    else if (arith == ARITH_F32_FMA) hipLaunchKernelGGL(finalize_ppm_kernel<ARITH_F32_FMA>, pixel_grid(p.map), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(finalize_ppm_kernel<ARITH_F32_LITERAL>, pixel_grid(p.map), dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace rtk
