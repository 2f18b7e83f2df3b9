/*
 * rt_oracle.cpp -- CPU restatement of the reference's per-pixel path-tracing loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls this
 * file; it exists so that tests, the smoke check and the benchmark's cpu_baseline leg
 * have an independent statement of what the reference computes.
 *
 * PARITY UNPINNED: the reference (ray-tracer/main.odin, one Odin file) holds no golden
 * vectors, no known-answer tests and no rendered image (out/image.ppm is absent from the
 * checkout), and no Odin compiler exists in this environment, so this restatement cannot
 * be checked against outputs of the reference itself.  It is instead cross-checked
 * against a second, independently written Python restatement (tests/pyref.py) and against
 * closed-form results.
 *
 * The random numbers do not come from Odin's core:math/rand (third-party, unpinned, its
 * default generator changed between Odin releases) but from the counter RNG defined below,
 * injected at exactly the reference's call sites and in the reference's draw order.
 *
 * Real = double, FMA = false : the reference's arithmetic as written (v3 :: [3]f64,
 *                              main.odin:19; LLVM does not contract a*b+c without fast-math).
 * Real = float,  FMA = false : the same operation order in f32.
 * Real = float,  FMA = true  : as above, but the ray/sphere test is the two-stage form of
 *                              ray_sphere_intersection_expanded below (candidates from an
 *                              expanded discriminant in fused multiply-adds, their roots from
 *                              the centre-to-line vector), the sphere a scattered segment
 *                              starts on is tested with c = 0
 *                              (ray_sphere_intersection_from_surface), a vector is divided by
 *                              a scalar as one reciprocal and three products, and sample
 *                              colours are summed exactly in Q32.32 with an f64 mean and
 *                              square root -- the arithmetic of the GPU throughput kernel.
 * RT_PRECISION_MIXED         : Real = double, FMA = false, with the GPU mixed mode's f32 filter
 *                              (mixed_filter below) in front of every ray/sphere test: a sphere
 *                              the filter drops is not tested.  The filter is meant to be
 *                              conservative, so the image must be that of RT_PRECISION_F64; the
 *                              run also evaluates the dropped spheres and counts every one that
 *                              main.odin:240-263 would have reported as hit (stats[6], must be 0).
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_mi355x.h" /* struct layouts only */

namespace {

/* ---------------------------------------------------------------- counter RNG
 * draw(seed, pixel, sample, bounce, j):
 *   K0 = mix(lo32(seed) ^ 0x9e3779b9)      K1 = mix(hi32(seed) ^ K0 ^ 0x85ebca6b)
 *   kp = mix(K0 ^ pixel)                   wp = mix(K1 + pixel)
 *   ks = mix(kp + 0x9e3779b9 * (sample+1))
 *   kb = mix(ks ^ (0x85ebca6b * (bounce+1)))
 *   u  = mix(kb + 0xc2b2ae35 * (j+1)) ^ wp
 * bounce 0 = the camera draws of a sample, bounce k+1 = the scatter draws at segment k.
 * uniform in [0,1): (u >> 8) * 2^-24 -- exact in f32 and f64, so both precisions see
 * the same random numbers.
 */
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
struct seed_keys { uint32_t k0, k1; };
inline seed_keys keys_from_seed(uint64_t seed) {
    seed_keys k;
    k.k0 = mix32(static_cast<uint32_t>(seed) ^ 0x9e3779b9u);
    k.k1 = mix32(static_cast<uint32_t>(seed >> 32) ^ k.k0 ^ 0x85ebca6bu);
    return k;
}
struct draw_stream { /* the draws of one (pixel, sample, bounce) */
    uint32_t kb, wp, j;
    draw_stream(seed_keys k, uint32_t pixel, uint32_t sample, uint32_t bounce) {
        uint32_t kp = mix32(k.k0 ^ pixel);
        wp = mix32(k.k1 + pixel);
        uint32_t ks = mix32(kp + 0x9e3779b9u * (sample + 1u));
        kb = mix32(ks ^ (0x85ebca6bu * (bounce + 1u)));
        j = 0;
    }
    uint32_t next_u32() { ++j; return mix32(kb + 0xc2b2ae35u * j) ^ wp; }
    template <class Real> Real next() { /* rand.float64(): uniform [0,1) */
        return static_cast<Real>(next_u32() >> 8) * static_cast<Real>(1.0 / 16777216.0);
    }
    /* rand.float64_range(lo, hi) of Odin core:math/rand: (hi-lo)*float64() + lo, and a
     * result that rounds up to hi is pulled back below it. */
    template <class Real> Real range(Real lo, Real hi) {
        Real val = (hi - lo) * next<Real>() + lo;
        if (val >= hi) val = std::max(lo, hi * (Real(1) - std::numeric_limits<Real>::epsilon()));
        return val;
    }
};

/* rejection loops are unbounded in the reference (main.odin:196, :216); with this RNG
 * 64 straight rejections have probability < 1e-20, after which the last candidate is
 * taken so that every loop provably ends. */
constexpr int REJECTION_CAP = 64;

/* ---------------------------------------------------------------- linalg  main.odin:19-63 */
template <class Real> struct vec3 { Real x, y, z; };
template <class Real> inline vec3<Real> operator+(vec3<Real> a, vec3<Real> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class Real> inline vec3<Real> operator-(vec3<Real> a, vec3<Real> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class Real> inline vec3<Real> operator-(vec3<Real> a) { return {-a.x, -a.y, -a.z}; }
template <class Real> inline vec3<Real> operator*(vec3<Real> a, vec3<Real> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class Real> inline vec3<Real> operator*(Real s, vec3<Real> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class Real> inline vec3<Real> operator*(vec3<Real> a, Real s) { return {a.x * s, a.y * s, a.z * s}; }
template <class Real> inline vec3<Real> operator/(vec3<Real> a, Real s) { return {a.x / s, a.y / s, a.z / s}; }

/* dot  main.odin:28-30: a.x*b.x + a.y*b.y + a.z*b.z, left to right */
template <class Real> inline Real dot(vec3<Real> a, vec3<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* cross  main.odin:32-38 */
template <class Real> inline vec3<Real> cross(vec3<Real> a, vec3<Real> b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
/* magnitude_squared / magnitude / normalize  main.odin:40-50 */
template <class Real> inline Real magnitude_squared(vec3<Real> v) { return dot(v, v); }
template <class Real> inline Real magnitude(vec3<Real> v) { return std::sqrt(magnitude_squared(v)); }
template <class Real> inline vec3<Real> normalize(vec3<Real> v) { return v / magnitude(v); }
/* GPU throughput arithmetic (FAST): one division, then three multiplications, wherever the
 * reference divides a vector by a scalar (main.odin:49, :200, :555) */
template <class Real, bool FAST> inline vec3<Real> scale_down(vec3<Real> v, Real s) {
    if (FAST) return v * (Real(1) / s);
    return v / s;
}
template <class Real, bool FAST> inline vec3<Real> normalize_as(vec3<Real> v) { return scale_down<Real, FAST>(v, magnitude(v)); }
/* is_near_zero  main.odin:52-57 */
template <class Real> inline bool is_near_zero(vec3<Real> v) {
    const Real EPSILON = Real(1e-8);
    return std::fabs(v.x) <= EPSILON && std::fabs(v.y) <= EPSILON && std::fabs(v.z) <= EPSILON;
}

/* ---------------------------------------------------------------- optics  main.odin:77-176 */
/* reflect  main.odin:77-80: vec - 2*dot(vec, normal)*normal */
template <class Real> inline vec3<Real> reflect(vec3<Real> v, vec3<Real> n) { return v - (Real(2) * dot(v, n)) * n; }
/* reflectance_at_normal_incidence  main.odin:111-115 */
template <class Real> inline Real reflectance_at_normal_incidence(Real rel) {
    Real r0 = (Real(1) - rel) / (Real(1) + rel);
    r0 *= r0;
    return r0;
}
/* reflectance_schlick_approximation  main.odin:119-122: r0 + (1-r0)*a*a*a*a*a */
template <class Real> inline Real reflectance_schlick_approximation(Real cos_i, Real r0) {
    Real a = Real(1) - cos_i;
    return r0 + (Real(1) - r0) * a * a * a * a * a;
}
/* refract_with_relative_refractive_index, arithmetic only  main.odin:168-173 */
template <class Real> inline vec3<Real> refract(vec3<Real> v, vec3<Real> n, Real rel) {
    Real cos_theta = std::min(dot(-v, n), Real(1));
    vec3<Real> perp = rel * (v + cos_theta * n);
    vec3<Real> par = (-std::sqrt(std::max(Real(1) - magnitude_squared(perp), Real(0)))) * n;
    return perp + par;
}

/* ---------------------------------------------------------------- random  main.odin:181-223 */
/* random_unit_vector  main.odin:194-203 (3 draws per candidate: x, y, z) */
template <class Real, bool FAST = false> inline vec3<Real> random_unit_vector(draw_stream &g) {
    const Real hi = Real(1) + Real(DBL_EPSILON); /* 1 + math.F64_EPSILON: main.odin:197 */
    vec3<Real> p{};
    Real lsq = 0;
    for (int it = 0; it < REJECTION_CAP; ++it) {
        p.x = g.range<Real>(Real(-1), hi);
        p.y = g.range<Real>(Real(-1), hi);
        p.z = g.range<Real>(Real(-1), hi);
        lsq = magnitude_squared(p);
        if (Real(0) < lsq && lsq <= Real(1)) break;
    }
    return scale_down<Real, FAST>(p, std::sqrt(lsq));
}
/* random_point_on_disk  main.odin:214-223 (2 draws per candidate; z = 0) */
template <class Real> inline vec3<Real> random_point_on_disk(draw_stream &g) {
    const Real hi = Real(1) + Real(DBL_EPSILON);
    vec3<Real> p{};
    for (int it = 0; it < REJECTION_CAP; ++it) {
        p.x = g.range<Real>(Real(-1), hi);
        p.y = g.range<Real>(Real(-1), hi);
        p.z = Real(0);
        if (magnitude_squared(p) <= Real(1)) break;
    }
    return p;
}

/* ---------------------------------------------------------------- raycast  main.odin:228-263 */
template <class Real> struct ray_t { vec3<Real> origin, direction; };
template <class Real> struct hit_record { vec3<Real> p, normal; bool front_face; };

/* ray_sphere_intersection  main.odin:240-263.  returns t in [t_min, t_max) or t_max. */
template <class Real>
inline Real ray_sphere_intersection(const ray_t<Real> &r, vec3<Real> c, Real radius, Real t_min, Real t_max) {
    Real t = t_max;
    vec3<Real> oc = c - r.origin;
    Real a = magnitude_squared(r.direction);
    Real h = dot(r.direction, oc);
    Real cc = magnitude_squared(oc) - radius * radius;
    Real disc = h * h - a * cc;
    if (disc < 0) return t;
    Real sqrtd = std::sqrt(disc);
    Real root = (h - sqrtd) / a;
    if (root < t_min || t_max <= root) {
        root = (h + sqrtd) / a;
        if (root < t_min || t_max <= root) return t;
    }
    t = root;
    return t;
}

/* Throughput arithmetic (RT_PRECISION_F32): the ray/sphere test in two stages.
 *
 * Stage 1, every sphere: the conservative filter below (mixed_filter) on the f32 ray.  It drops a
 * sphere only when its discriminant is negative by more than every rounding error, or when the
 * line certainly meets it only behind an origin certainly outside it, so what stage 2 decides
 * for the spheres it keeps is all that reaches the image.
 *
 * Stage 2, kept spheres only: the roots from the vector l that runs from the centre to the
 * line's closest point, whose length is of the size of the radius, so that the rounding error
 * scales with the sphere and not with its distance:
 *     oc = center - o,   q = fma(ocz, dz, fma(ocy, dy, ocx*dx)) * (1/a),
 *     l  = fma(-q, d, oc) per component,   D = r*r - fma(lz, lz, fma(ly, ly, lx*lx)),
 *     roots q -+ sqrt(D * (1/a)),
 * which is main.odin:249-258 with disc = a*D; the range tests and the order of the two roots
 * (main.odin:250-261) are unchanged.  r*r is the f32 product of the f32 radius. */
struct expanded_ray { float dx, dy, dz, a, inv_a; vec3<float> o; };
inline expanded_ray expand_ray(const ray_t<float> &r) {
    expanded_ray e;
    e.dx = r.direction.x; e.dy = r.direction.y; e.dz = r.direction.z;
    e.a = dot(r.direction, r.direction);
    e.inv_a = 1.0f / e.a; /* one reciprocal per segment instead of a division per root */
    e.o = r.origin;
    return e;
}
/* The candidate filter of the GPU throughput and mixed modes: seven fused f32 operations per sphere on the
 * f32 ray (mixed mode: the f64 ray rounded to f32), with an error margin, so that no sphere whose exact
 * discriminant (main.odin:249) is non-negative -- in particular none the f64 test of main.odin:245-261
 * reports as hit -- is ever dropped.  The ray is taken as a unit direction e = d/|d| and the foot point
 * f = o - (o.e) e of the coordinate origin on its line; then
 *     r^2 - dist(center, line)^2 = (c.e)^2 - K + 2 c.f - |f|^2,   K = |center|^2 - r^2
 * (= disc / a of main.odin:249), evaluated as
 *     h = fma(cz, ez, fma(cy, ey, cx*ex)),   w = fma(cz, 2fz, fma(cy, 2fy, fma(cx, 2fx, fma(h, h, -K)))),
 * and the sphere is kept unless w < |f|^2.  With u = 2^-24 and S = |center| + |o| + r, w - |f|^2
 * differs from the f64 value by at most 43 u S^2: 18 from rounding d, o, center, r to f32 and from the
 * roundings of e and f (the line moves by at most 8 u S near a sphere it meets), 10 because |e|^2 - 1
 * and f.e are not exactly zero (at most 9 u and 13 u |o|), 10 from evaluating h and w, 4 from |f|^2,
 * 1 from the difference.  S^2 <= 3(|center|^2 + r^2 + |o|^2), so K is lowered by 2^-16 (|center|^2 +
 * r^2) (the record holds -K', rounded up) and |f|^2 by 2^-16 |o|^2: 85 u S^2 in all, 42 to spare.
 * The spare part makes a second decision certain: hs = h - o.e, the center's parameter relative to
 * the origin per unit direction, is off by at most 12 u S, so hs < -behind with behind = 2^-19
 * (reach + |o|_1) >= 32 u S is certainly negative (|.|_1 the sum of absolute components, an upper bound
 * of the length that needs no square root), and hs^2 > w - thr then makes c = hs^2 - disc/a of
 * main.odin:248 certainly positive (24 u S^2 for the error of hs^2): the line meets the sphere only
 * behind an origin outside it, both roots are negative, the sphere is dropped.
 * The sphere a scattered segment starts on is dropped under the same condition on hs when its c
 * (main.odin:248) is known to be tiny: the origin p = o' + t d' was a root of that sphere's
 * equation, so |c| <= 64 * 2^-53 (reach + |o'|)^2 =: B (o' the previous origin; 21 units from the
 * root and the hit point, 4 from evaluating c), and with |h| >= |d| behind/2 the roots of
 * main.odin:253-257 are <= 0 (c >= 0) or the far one is c/(h - sqrt(disc)) <= B/(|d| behind), which is
 * below t_min when B < 9e-4 |d| behind.  The bound
 * needs normal f32 numbers throughout: spheres with 2^-20 <= |r| and |center| + |r| <= 2^20 (else
 * the filter is not used at all), rays with 2^-40 <= |d|^2 <= 2^40 and |o|^2 <= 2^40 (else the ray keeps
 * every sphere). */
struct filter_ray { vec3<float> e, f2; float s_o, thr, nd, behind, reach_o; bool in_range; };
inline float sphere_constant_lowered(float cx, float cy, float cz, float radius) {
    const double c2 = static_cast<double>(cx) * cx + static_cast<double>(cy) * cy + static_cast<double>(cz) * cz, r2 = static_cast<double>(radius) * radius;
    const double lowered = (c2 - r2) - (c2 + r2) * (1.0 / 65536.0);
    float kn = static_cast<float>(-lowered); /* -K' */
    if (static_cast<double>(kn) < -lowered) kn = std::nextafterf(kn, INFINITY);
    return kn;
}
inline filter_ray make_filter_ray(const ray_t<double> &r, float reach) {
    filter_ray f;
    const vec3<float> fo{static_cast<float>(r.origin.x), static_cast<float>(r.origin.y), static_cast<float>(r.origin.z)};
    const vec3<float> fd{static_cast<float>(r.direction.x), static_cast<float>(r.direction.y), static_cast<float>(r.direction.z)};
    const float fa = dot(fd, fd), o_o = dot(fo, fo);
    f.nd = std::sqrt(fa);
    const float inv_nd = 1.0f / f.nd;
    f.e = {fd.x * inv_nd, fd.y * inv_nd, fd.z * inv_nd};
    f.s_o = dot(fo, f.e);
    const vec3<float> ft{std::fmaf(-f.s_o, f.e.x, fo.x), std::fmaf(-f.s_o, f.e.y, fo.y), std::fmaf(-f.s_o, f.e.z, fo.z)};
    f.thr = std::fmaf(-1.52587890625e-05f, o_o, dot(ft, ft));
    f.f2 = {ft.x + ft.x, ft.y + ft.y, ft.z + ft.z};
    f.reach_o = reach + (std::fabs(fo.x) + std::fabs(fo.y) + std::fabs(fo.z));
    f.behind = 1.9073486328125e-06f * f.reach_o;
    f.in_range = (fa >= 9.094947017729282e-13f) && (fa <= 1.099511627776e12f) && (o_o <= 1.099511627776e12f) && (reach <= 1048576.0f);
    if (!f.in_range) f.reach_o = FLT_MAX;
    return f;
}
/* the filter's values for one sphere: h = c.e and w (see above); Kn = -K' */
inline void mixed_filter_values(const filter_ray &f, vec3<float> c, float Kn, float &h, float &w) {
    h = std::fmaf(c.z, f.e.z, std::fmaf(c.y, f.e.y, c.x * f.e.x));
    w = std::fmaf(c.z, f.f2.z, std::fmaf(c.y, f.f2.y, std::fmaf(c.x, f.f2.x, std::fmaf(h, h, Kn))));
}
/* on_start: the sphere is the one the segment starts on and its c is known to be below 9e-4 |d| behind */
inline bool mixed_filter(const filter_ray &f, vec3<float> c, float Kn, bool on_start = false) {
    if (!f.in_range) return true;
    float h, w;
    mixed_filter_values(f, c, Kn, h, w);
    if (w < f.thr) return false;
    const float hs = h - f.s_o;
    if (hs < -f.behind && (std::fmaf(hs, hs, -(w - f.thr)) > 0 || on_start)) return false;
    return true;
}

inline float ray_sphere_intersection_expanded(const expanded_ray &e, const filter_ray &f, vec3<float> c, float Kn, float radius, float t_min, float t_max) {
    /* stage 1 */
    if (!mixed_filter(f, c, Kn)) return t_max;
    /* stage 2 */
    const float ocx = c.x - e.o.x, ocy = c.y - e.o.y, ocz = c.z - e.o.z;
    const float q = std::fmaf(ocz, e.dz, std::fmaf(ocy, e.dy, ocx * e.dx)) * e.inv_a;
    const float lx = std::fmaf(-q, e.dx, ocx), ly = std::fmaf(-q, e.dy, ocy), lz = std::fmaf(-q, e.dz, ocz);
    const float D = radius * radius - std::fmaf(lz, lz, std::fmaf(ly, ly, lx * lx));
    if (!(D >= 0)) return t_max;
    const float s = std::sqrt(D * e.inv_a);
    float root = q - s;
    if (root < t_min || t_max <= root) {
        root = q + s;
        if (root < t_min || t_max <= root) return t_max;
    }
    return root;
}

/* Throughput arithmetic, the sphere a scattered segment starts on.  Its origin is the hit point of
 * the previous segment, so c = |center - o|^2 - r^2 is zero up to the rounding of that hit
 * point; in f64 the reference's expression sees |c| ~ 1e-15 and its near root falls below t_min,
 * in f32 the rounding of c is large enough to produce spurious hits at t just above t_min for
 * short or grazing directions.  The test therefore takes c = 0 for this one sphere: the roots of
 * main.odin:249-258 become 0 and 2h/a, of which only the second can lie in [t_min, t_max), and
 * only for h > 0 (a segment that runs through the inside of the sphere).  q = h/a is formed as in
 * stage 2 of ray_sphere_intersection_expanded.  trace_sample tests this sphere first and leaves
 * it out of the scan over the list. */
inline float ray_sphere_intersection_from_surface(const expanded_ray &e, vec3<float> c, float t_min, float t_max) {
    const float ocx = c.x - e.o.x, ocy = c.y - e.o.y, ocz = c.z - e.o.z;
    const float q = std::fmaf(ocz, e.dz, std::fmaf(ocy, e.dy, ocx * e.dx)) * e.inv_a;
    const float root = q + q;
    if (q > 0 && root >= t_min && root < t_max) return root;
    return t_max;
}

/* ---------------------------------------------------------------- materials  main.odin:268-410 */
template <class Real> struct material_t { int64_t type; vec3<Real> albedo; Real param1; };

template <class Real> struct scatter_result { ray_t<Real> ray_out; vec3<Real> attenuation; bool ok; };

/* lambertian_proc  main.odin:292-312 */
template <class Real, bool FAST = false>
inline scatter_result<Real> lambertian_proc(const material_t<Real> &m, const ray_t<Real> &, const hit_record<Real> &hit, draw_stream &g) {
    vec3<Real> dir = hit.normal + random_unit_vector<Real, FAST>(g);
    if (is_near_zero(dir)) dir = hit.normal;
    return {{hit.p, dir}, m.albedo, true};
}
/* reflectance_fresnel  main.odin:84-109 */
template <class Real> inline Real reflectance_fresnel(Real cos_i, Real sin_i, Real n1, Real n2) {
    Real subexp = (n1 / n2) * sin_i;
    subexp = Real(1) - subexp * subexp;
    subexp = std::sqrt(subexp);
    Real x1 = n1 * cos_i, x2 = n2 * subexp;
    Real r_s = (x1 - x2);
    r_s /= (x1 + x2);
    r_s *= r_s;
    x1 = n1 * subexp; x2 = n2 * cos_i;
    Real r_p = (x1 - x2);
    r_p /= (x1 + x2);
    r_p *= r_p;
    return (r_s + r_p) * Real(0.5);
}
/* reflectance_schlick_lazanyi_approximation  main.odin:132-134.  LITERAL: as the reference's text reads, the Schlick
 * term is called with (r0, cos_i) where the procedure declares (cos_i, r0); otherwise in declared order, as the cited
 * paper has it.  math.pow is core:math (third party, not in the checkout); for the only exponent used, 6, it is
 * taken as ((x*x)*x)^2. */
template <class Real, bool LITERAL> inline Real reflectance_schlick_lazanyi_approximation(Real cos_i, Real r0, Real a) {
    const Real x = Real(1) - cos_i;
    const Real x2 = x * x, x3 = x2 * x;
    const Real schlick = LITERAL ? reflectance_schlick_approximation(r0, cos_i) : reflectance_schlick_approximation(cos_i, r0);
    return schlick - a * cos_i * (x3 * x3);
}
/* reflectance_hoffman_approximation  main.odin:141-144.  LITERAL: 823543 / 46656 and 49 / 6 are divisions of untyped
 * integer constants, 17 and 8; otherwise the real quotients 7^7/6^6 and 49/6 of the cited paper. */
template <class Real, bool LITERAL> inline Real reflectance_hoffman_approximation(Real cos_i, Real r0, Real h) {
    const Real a = (LITERAL ? Real(17) : static_cast<Real>(823543.0 / 46656.0)) * (r0 - h) + (LITERAL ? Real(8) : static_cast<Real>(49.0 / 6.0)) * (Real(1) - r0);
    return reflectance_schlick_lazanyi_approximation<Real, LITERAL>(cos_i, r0, a);
}
/* metallic_proc  main.odin:315-353.  kind < 0: the USE_METAL_FRESNEL = false branch (:348-350); otherwise that flag
 * set with METAL_FRESNEL_KIND = kind: 2 is what the reference selects (:341-344), 0 (:334-335) and 1 (:337-339) are
 * its alternatives.  1.0/HARDCODED_REFRACTIVE_INDEX is a constant expression: the exact quotient rounded once. */
template <class Real, bool FAST = false>
inline scatter_result<Real> metallic_proc(const material_t<Real> &m, const ray_t<Real> &ray_in, const hit_record<Real> &hit, draw_stream &g, int kind) {
    Real fuzz = m.param1;
    vec3<Real> dir = normalize_as<Real, FAST>(reflect(ray_in.direction, hit.normal));
    dir = dir + fuzz * random_unit_vector<Real, FAST>(g);
    vec3<Real> attenuation = m.albedo;
    if (kind >= 0) {
        Real cos_theta = dot(-normalize_as<Real, FAST>(ray_in.direction), hit.normal);
        cos_theta = std::min(cos_theta, Real(1));
        Real reflection_factor;
        if (kind == 2) {
            const Real sin_theta = std::sqrt(Real(1) - cos_theta * cos_theta);
            reflection_factor = reflectance_fresnel<Real>(cos_theta, sin_theta, Real(1), static_cast<Real>(1.27035));
        } else {
            const Real r0 = reflectance_at_normal_incidence(static_cast<Real>(0.7871846341559413)); /* 1.0/1.27035 */
            reflection_factor = kind == 0 ? reflectance_schlick_approximation(cos_theta, r0)
                              : kind == 1 ? reflectance_hoffman_approximation<Real, true>(cos_theta, r0, Real(0.5))    /* the reference's text */
                                          : reflectance_hoffman_approximation<Real, false>(cos_theta, r0, Real(0.5)); /* kind 3: the cited paper's form */
        }
        const vec3<Real> one{Real(1), Real(1), Real(1)};
        attenuation = m.albedo * (Real(1) - reflection_factor) + one * reflection_factor; /* math.lerp(albedo, 1, t) */
    }
    return {{hit.p, dir}, attenuation, dot(dir, hit.normal) > Real(0)};
}
/* dielectric_proc  main.odin:356-400 */
template <class Real, bool FAST = false>
inline scatter_result<Real> dielectric_proc(const material_t<Real> &m, const ray_t<Real> &ray_in, const hit_record<Real> &hit, draw_stream &g) {
    Real ri = m.param1;
    Real src = ri >= Real(1) ? Real(1) : Real(1) / ri;
    Real dst = ri >= Real(1) ? ri : Real(1);
    if (!hit.front_face) std::swap(src, dst);
    Real rel = src / dst;
    vec3<Real> unit = normalize_as<Real, FAST>(ray_in.direction);
    Real cos_theta = dot(-unit, hit.normal);
    cos_theta = std::min(cos_theta, Real(1));
    Real sin2 = Real(1) - cos_theta * cos_theta;
    bool must_reflect = (rel * rel * sin2) > Real(1);
    vec3<Real> dir;
    /* `||` short-circuits: the uniform is drawn only when !must_reflect (main.odin:390) */
    if (must_reflect || reflectance_schlick_approximation(cos_theta, reflectance_at_normal_incidence(rel)) > g.next<Real>())
        dir = reflect(unit, hit.normal);
    else
        dir = refract(unit, hit.normal, rel);
    return {{hit.p, dir}, {Real(1), Real(1), Real(1)}, true};
}
/* material_scatter  main.odin:403-410 */
template <class Real, bool FAST = false>
inline scatter_result<Real> material_scatter(const material_t<Real> &m, const ray_t<Real> &ray_in, const hit_record<Real> &hit, draw_stream &g) {
    switch (m.type) {
    case RT_MATERIAL_LAMBERTIAN: return lambertian_proc<Real, FAST>(m, ray_in, hit, g);
    case RT_MATERIAL_METALLIC: return metallic_proc<Real, FAST>(m, ray_in, hit, g, -1);
    case RT_MATERIAL_METALLIC_FRESNEL: return metallic_proc<Real, FAST>(m, ray_in, hit, g, 2);
    case RT_MATERIAL_METALLIC_SCHLICK: return metallic_proc<Real, FAST>(m, ray_in, hit, g, 0);
    case RT_MATERIAL_METALLIC_HOFFMAN: return metallic_proc<Real, FAST>(m, ray_in, hit, g, 1);
    case RT_MATERIAL_METALLIC_HOFFMAN_PAPER: return metallic_proc<Real, FAST>(m, ray_in, hit, g, 3);
    default: return dielectric_proc<Real, FAST>(m, ray_in, hit, g);
    }
}

/* background_color  main.odin:412-418; math.lerp(a, b, t) = a*(1-t) + b*t */
template <class Real> inline vec3<Real> background_color(const ray_t<Real> &r) {
    vec3<Real> a{Real(1), Real(1), Real(1)};
    vec3<Real> b{Real(0.5), Real(0.7), Real(1.0)};
    Real t = Real(0.5) * (r.direction.y + Real(1));
    return a * (Real(1) - t) + b * t;
}

/* ---------------------------------------------------------------- the world in Real */
template <class Real> struct sphere_t { vec3<Real> center; Real radius; int64_t material_index; float inv_radius; /* 1/r: throughput arithmetic only */
                                        vec3<float> center32; float Km; /* the filter's record: f32 center and -K' */ };
template <class Real> struct camera_t {
    vec3<Real> position, du, dv, top_left, dof_u, dof_v;
    Real dof_angle;
};
template <class Real> inline vec3<Real> cvt(const rt_v3 &v) { return {static_cast<Real>(v.x), static_cast<Real>(v.y), static_cast<Real>(v.z)}; }

template <class Real> struct scene_t {
    std::vector<sphere_t<Real>> spheres;
    std::vector<material_t<Real>> materials;
    camera_t<Real> cam;
    int64_t max_ray_bounces;
    float filter_reach = 0.0f;  /* mixed mode: max |center| + |r| of the f32 records, rounded up */
    bool filter_ok = true;      /* every sphere inside the range the filter's bound needs */
    scene_t(const rt_camera_render_data *c, const rt_sphere *s, int64_t ns, const rt_material *m, int64_t nm) {
        spheres.resize(ns);
        double reach = 0.0;
        for (int64_t i = 0; i < ns; ++i) {
            const float cx = static_cast<float>(s[i].center.x), cy = static_cast<float>(s[i].center.y), cz = static_cast<float>(s[i].center.z), rf = static_cast<float>(s[i].radius);
            spheres[i] = {cvt<Real>(s[i].center), static_cast<Real>(s[i].radius), s[i].material_index,
                          1.0f / rf, {cx, cy, cz}, sphere_constant_lowered(cx, cy, cz, rf)};
            const double far = std::sqrt(static_cast<double>(cx) * cx + static_cast<double>(cy) * cy + static_cast<double>(cz) * cz) + std::fabs(static_cast<double>(rf));
            reach = std::max(reach, far);
            if (!(std::fabs(static_cast<double>(rf)) >= 1.0 / 1048576.0) || !(far <= 1048576.0)) filter_ok = false;
        }
        filter_reach = static_cast<float>(reach);
        if (static_cast<double>(filter_reach) < reach) filter_reach = std::nextafterf(filter_reach, INFINITY);
        materials.resize(nm);
        for (int64_t i = 0; i < nm; ++i) materials[i] = {m[i].type, cvt<Real>(m[i].albedo), static_cast<Real>(m[i].param1)};
        cam = {cvt<Real>(c->position), cvt<Real>(c->pixel_delta_u), cvt<Real>(c->pixel_delta_v),
               cvt<Real>(c->view_plane_top_left_pixel_center), cvt<Real>(c->dof_disk_u), cvt<Real>(c->dof_disk_v),
               static_cast<Real>(c->depth_of_field_angle)};
        max_ray_bounces = c->max_ray_bounces;
    }
};

/* one sample of one pixel: main.odin:518-575.  returns sample_color; adds executed
 * bounce-loop iterations to *segments. */
struct filter_counters { uint64_t kept = 0, wrongly_dropped = 0; };
template <class Real, bool FMA, bool MIXED = false>
inline vec3<Real> trace_sample(const scene_t<Real> &sc, seed_keys keys, int64_t u, int64_t v, int64_t width, uint32_t sample, uint64_t *segments, filter_counters *fc = nullptr) {
    const camera_t<Real> &cam = sc.cam;
    const uint32_t pixel = static_cast<uint32_t>(u + v * width);
    draw_stream cg(keys, pixel, sample, 0);
    /* random_v2_range(-0.5, 0.5)  main.odin:519, :190-192 */
    Real ox = cg.range<Real>(Real(-0.5), Real(0.5));
    Real oy = cg.range<Real>(Real(-0.5), Real(0.5));
    vec3<Real> pixel_sample_position = cam.top_left + (static_cast<Real>(u) + ox) * cam.du + (static_cast<Real>(v) + oy) * cam.dv;
    vec3<Real> ray_origin = cam.position;
    if (cam.dof_angle > Real(0)) {
        vec3<Real> dof_sample = random_point_on_disk<Real>(cg);
        ray_origin = ray_origin + ((dof_sample.x * cam.dof_u) + (dof_sample.y * cam.dof_v));
    }
    ray_t<Real> ray{ray_origin, pixel_sample_position - ray_origin};
    vec3<Real> sample_color{Real(1), Real(1), Real(1)};
    const int64_t n = static_cast<int64_t>(sc.spheres.size());
    int64_t start_sphere = -1; /* throughput and mixed arithmetic: the sphere this segment's origin lies on (none for a camera ray) */
    float start_cc_bound = 0.0f, next_cc_bound = 0.0f; /* mixed mode: bound on |c| of main.odin:248 for that sphere */
    for (int64_t bounce = 0; bounce <= sc.max_ray_bounces; ++bounce) { /* 0..=max  main.odin:535 */
        ++*segments;
        Real closest_t = std::numeric_limits<Real>::max();
        int64_t hit_index = -1;
        const Real SHADOW_ACNE_RAY_OFFSET = Real(0.001);
        if constexpr (FMA) {
            const expanded_ray e = expand_ray(ray);
            const filter_ray f = make_filter_ray(ray_t<double>{{ray.origin.x, ray.origin.y, ray.origin.z}, {ray.direction.x, ray.direction.y, ray.direction.z}},
                                                 sc.filter_ok ? sc.filter_reach : FLT_MAX);
            if (start_sphere >= 0) {
                Real t = ray_sphere_intersection_from_surface(e, sc.spheres[start_sphere].center, SHADOW_ACNE_RAY_OFFSET, closest_t);
                if (t < closest_t) { closest_t = t; hit_index = start_sphere; }
            }
            for (int64_t i = 0; i < n; ++i) {
                if (i == start_sphere) continue;
                Real t = ray_sphere_intersection_expanded(e, f, sc.spheres[i].center32, sc.spheres[i].Km, sc.spheres[i].radius, SHADOW_ACNE_RAY_OFFSET, closest_t);
                if (t < closest_t) { closest_t = t; hit_index = i; }
            }
        } else if constexpr (MIXED) {
            const filter_ray f = make_filter_ray(ray, sc.filter_reach);
            const bool skip_start = start_sphere >= 0 && f.in_range && start_cc_bound < 9.0e-4f * (f.nd * f.behind);
            next_cc_bound = 7.2e-15f * f.reach_o * f.reach_o; /* 64 * 2^-53 (reach + |o|)^2, for the segment that follows */
            for (int64_t i = 0; i < n; ++i) {
                const bool keep = !sc.filter_ok || mixed_filter(f, sc.spheres[i].center32, sc.spheres[i].Km, skip_start && i == start_sphere);
                Real t = ray_sphere_intersection<Real>(ray, sc.spheres[i].center, sc.spheres[i].radius, SHADOW_ACNE_RAY_OFFSET, closest_t);
                if (!keep) { if (t < closest_t) ++fc->wrongly_dropped; continue; } /* evaluated only to audit the filter */
                ++fc->kept;
                if (t < closest_t) { closest_t = t; hit_index = i; }
            }
        } else {
            for (int64_t i = 0; i < n; ++i) {
                Real t = ray_sphere_intersection<Real>(ray, sc.spheres[i].center, sc.spheres[i].radius, SHADOW_ACNE_RAY_OFFSET, closest_t);
                if (t < closest_t) { closest_t = t; hit_index = i; }
            }
        }
        if (hit_index >= 0) {
            const sphere_t<Real> &sp = sc.spheres[hit_index];
            const material_t<Real> &mat = sc.materials[sp.material_index];
            hit_record<Real> hit;
            hit.p = ray.origin + closest_t * ray.direction;
            if constexpr (FMA) hit.normal = (hit.p - sp.center) * static_cast<Real>(sp.inv_radius);
            else hit.normal = (hit.p - sp.center) / sp.radius;
            hit.front_face = dot(ray.direction, hit.normal) < Real(0);
            hit.normal = hit.front_face ? hit.normal : -hit.normal;
            draw_stream sg(keys, pixel, sample, static_cast<uint32_t>(bounce) + 1u);
            scatter_result<Real> s = material_scatter<Real, FMA>(mat, ray, hit, sg);
            if (s.ok) {
                start_sphere = hit_index;
                start_cc_bound = next_cc_bound;
                ray = s.ray_out;
                sample_color = sample_color * s.attenuation;
            } else {
                sample_color = {Real(0), Real(0), Real(0)};
                break;
            }
        } else {
            ray.direction = normalize_as<Real, FMA>(ray.direction);
            sample_color = sample_color * background_color(ray);
            break;
        }
    }
    return sample_color;
}

struct render_args {
    int64_t width, height;
    rt_rect region;
    const rt_camera_render_data *cam;
    const rt_sphere *spheres; int64_t n_spheres;
    const rt_material *materials; int64_t n_materials;
    uint64_t seed;
    int64_t sample_begin, sample_count, sample_chunks;
    double *image, *linear, *accum;
    uint64_t *stats;
    int32_t threads;
    uint64_t *words = nullptr; /* RT_PRECISION_MIXED_FIXED: optional width*height*3*2 accumulator words {a, b} */
};

/* Q32.32 fixed point of a non-negative f32, truncated below 2^-32 (NaN and negatives -> 0):
 * the accumulator format of the GPU throughput kernel, whose lanes finish samples in no
 * fixed order and therefore need a sum that does not depend on the order of addition. */
inline uint64_t to_q32_32(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFF00000000ull;
    const uint32_t hi = static_cast<uint32_t>(x);
    const float fr = x - static_cast<float>(hi);
    const uint32_t lo = static_cast<uint32_t>(fr * 4294967296.0f);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
inline double q32_32_to_f64(uint64_t v) {
    return static_cast<double>(static_cast<uint32_t>(v >> 32)) * 4294967296.0 + static_cast<double>(static_cast<uint32_t>(v));
}

/* Q32.64 fixed point of a non-negative f64 as two words (the accumulator format of RT_PRECISION_MIXED_FIXED):
 * a = floor(x * 2^32), b = the next 32 bits; truncated below 2^-64, NaN and negatives -> 0, x >= 2^32 saturates. */
inline void to_q32_64(double x, uint64_t &a, uint64_t &b) {
    a = 0; b = 0;
    if (!(x > 0.0)) return;
    if (x >= 4294967296.0) { a = ~0ull; b = 0xFFFFFFFFull; return; }
    const uint32_t hi = static_cast<uint32_t>(x);
    const double t1 = (x - static_cast<double>(hi)) * 4294967296.0;
    const uint32_t mid = static_cast<uint32_t>(t1);
    const double t2 = (t1 - static_cast<double>(mid)) * 4294967296.0;
    a = (static_cast<uint64_t>(hi) << 32) | mid;
    b = static_cast<uint64_t>(static_cast<uint32_t>(t2));
}
/* a * 2^-32 + b * 2^-64 rounded to f64 once: the 128-bit integer a * 2^32 + b through unsigned __int128, whose
 * conversion would be the compiler's business; here its top 64 bits with a sticky lowest bit, converted (one
 * rounding) and scaled by a power of two. */
inline double q32_64_to_f64(uint64_t a, uint64_t b) {
    const unsigned __int128 t = (static_cast<unsigned __int128>(a) << 32) + b;
    const uint64_t hi = static_cast<uint64_t>(t >> 64), lo = static_cast<uint64_t>(t);
    if (hi == 0) return static_cast<double>(lo) * 5.421010862427522e-20; /* 2^-64 */
    const int shift = __builtin_clzll(hi);
    const unsigned __int128 n = t << shift;
    uint64_t top = static_cast<uint64_t>(n >> 64);
    if (static_cast<uint64_t>(n) != 0) top |= 1;
    return std::ldexp(static_cast<double>(top), -shift);
}

/* render_region  main.odin:510-591 over global sample indices
 * [sample_begin, sample_begin + sample_count).
 * QACC = false: pixel_color is a Real sum in sample order, the mean and the gamma square root
 *   are taken in Real (the reference's loop).  With sample_chunks = c > 1 the samples are summed
 *   in c consecutive groups of ceil(count/c) and the group sums are then added in order (the
 *   order in which the GPU combines its per-chunk partial sums).
 * QACC = true (f32 throughput arithmetic): each f32 sample colour is converted to Q32.32 and
 *   summed exactly in u64; mean and square root are taken in f64. */
template <class Real, bool FMA, bool QACC, bool MIXED = false, bool FIXED = false> void render(const render_args &A) {
    scene_t<Real> sc(A.cam, A.spheres, A.n_spheres, A.materials, A.n_materials);
    const seed_keys keys = keys_from_seed(A.seed);
    const Real contribution = Real(1) / static_cast<Real>(A.sample_count); /* main.odin:512 */
    const double contribution64 = 1.0 / static_cast<double>(A.sample_count);
    const int64_t chunks = std::max<int64_t>(1, A.sample_chunks);
    const int64_t per = (A.sample_count + chunks - 1) / chunks;
    std::atomic<int64_t> next_row{A.region.min_y};
    std::atomic<uint64_t> total_segments{0}, total_kept{0}, total_wrong{0};
    auto work = [&]() {
        uint64_t segments = 0;
        filter_counters fc;
        for (;;) {
            int64_t v = next_row.fetch_add(1);
            if (v >= A.region.one_past_max_y) break;
            for (int64_t u = A.region.min_x; u < A.region.one_past_max_x; ++u) {
                const int64_t index = u + v * A.width;
                if (QACC) {
                    uint64_t q[3] = {0, 0, 0};
                    for (int64_t s = 0; s < A.sample_count; ++s) {
                        vec3<Real> c = trace_sample<Real, FMA>(sc, keys, u, v, A.width, static_cast<uint32_t>(A.sample_begin + s), &segments);
                        q[0] += to_q32_32(static_cast<float>(c.x)); q[1] += to_q32_32(static_cast<float>(c.y)); q[2] += to_q32_32(static_cast<float>(c.z));
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double mean = (q32_32_to_f64(q[k]) * (1.0 / 4294967296.0)) * contribution64;
                        if (A.accum) A.accum[index * 3 + k] = static_cast<double>(q[k]); /* exact below 2^53 */
                        if (A.linear) A.linear[index * 3 + k] = mean;
                        if (A.image) A.image[index * 3 + k] = std::sqrt(mean);
                    }
                    continue;
                }
                if (FIXED) {
                    /* RT_PRECISION_MIXED_FIXED: the mixed mode's f64 sample colours, added exactly in Q32.64; the sum
                     * is rounded to f64 once, then the mean and the square root as main.odin:577-583 */
                    uint64_t qa[3] = {0, 0, 0}, qb[3] = {0, 0, 0};
                    for (int64_t s = 0; s < A.sample_count; ++s) {
                        vec3<Real> c = trace_sample<Real, FMA, MIXED>(sc, keys, u, v, A.width, static_cast<uint32_t>(A.sample_begin + s), &segments, &fc);
                        const double ch[3] = {static_cast<double>(c.x), static_cast<double>(c.y), static_cast<double>(c.z)};
                        for (int k = 0; k < 3; ++k) { uint64_t a, b; to_q32_64(ch[k], a, b); qa[k] += a; qb[k] += b; }
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double sum = q32_64_to_f64(qa[k], qb[k]);
                        const double mean = sum * contribution64;
                        if (A.accum) A.accum[index * 3 + k] = sum;
                        if (A.words) { A.words[(index * 3 + k) * 2] = qa[k]; A.words[(index * 3 + k) * 2 + 1] = qb[k]; }
                        if (A.linear) A.linear[index * 3 + k] = mean;
                        if (A.image) A.image[index * 3 + k] = std::sqrt(mean);
                    }
                    continue;
                }
                vec3<Real> pixel_color{0, 0, 0};
                for (int64_t c = 0; c < chunks; ++c) {
                    vec3<Real> part{0, 0, 0};
                    int64_t lo = c * per, hi = std::min(A.sample_count, (c + 1) * per);
                    for (int64_t s = lo; s < hi; ++s)
                        part = part + trace_sample<Real, FMA, MIXED>(sc, keys, u, v, A.width, static_cast<uint32_t>(A.sample_begin + s), &segments, &fc);
                    pixel_color = (chunks == 1) ? part : pixel_color + part;
                }
                if (A.accum) { A.accum[index * 3 + 0] = pixel_color.x; A.accum[index * 3 + 1] = pixel_color.y; A.accum[index * 3 + 2] = pixel_color.z; }
                pixel_color = pixel_color * contribution; /* main.odin:577 */
                if (A.linear) { A.linear[index * 3 + 0] = pixel_color.x; A.linear[index * 3 + 1] = pixel_color.y; A.linear[index * 3 + 2] = pixel_color.z; }
                /* linear to gamma 2  main.odin:581-583 */
                if (A.image) {
                    A.image[index * 3 + 0] = std::sqrt(pixel_color.x);
                    A.image[index * 3 + 1] = std::sqrt(pixel_color.y);
                    A.image[index * 3 + 2] = std::sqrt(pixel_color.z);
                }
            }
        }
        total_segments += segments;
        total_kept += fc.kept;
        total_wrong += fc.wrongly_dropped;
    };
    int32_t nt = std::max(1, A.threads);
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int32_t i = 0; i < nt; ++i) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (A.stats) {
        uint64_t npix = static_cast<uint64_t>(std::max<int64_t>(0, A.region.one_past_max_x - A.region.min_x)) *
                        static_cast<uint64_t>(std::max<int64_t>(0, A.region.one_past_max_y - A.region.min_y));
        A.stats[RT_STAT_SEGMENTS] = total_segments.load();
        A.stats[RT_STAT_SPHERE_TESTS] = total_segments.load() * static_cast<uint64_t>(A.n_spheres);
        A.stats[RT_STAT_SAMPLES] = npix * static_cast<uint64_t>(A.sample_count);
        if (MIXED) { A.stats[5] = total_kept.load(); A.stats[6] = total_wrong.load(); } /* spheres the filter kept / dropped although hit */
    }
}

} // namespace

extern "C" {

uint32_t oracle_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t draw) {
    draw_stream g(keys_from_seed(seed), pixel, sample, bounce);
    uint32_t u = 0;
    for (uint32_t i = 0; i <= draw; ++i) u = g.next_u32();
    return u;
}

/* precision: RT_PRECISION_F32 (fma scan), RT_PRECISION_F64, RT_PRECISION_F32_LITERAL.
 * image / linear / accum: optional width*height*3 doubles (gamma image, sample mean, raw sum).
 * stats: optional RT_STATS_COUNT u64.  threads: worker threads over rows. */
int32_t oracle_render_region(int32_t precision, double *image, double *linear, double *accum,
                             int64_t width, int64_t height, rt_rect region,
                             const rt_camera_render_data *cam,
                             const rt_sphere *spheres, int64_t n_spheres,
                             const rt_material *materials, int64_t n_materials,
                             uint64_t seed, int64_t sample_begin, int64_t sample_count, int64_t sample_chunks,
                             uint64_t *stats, int32_t threads) {
    render_args A{width, height, region, cam, spheres, n_spheres, materials, n_materials, seed,
                  sample_begin, sample_count, sample_chunks, image, linear, accum, stats, threads};
    switch (precision) {
    case RT_PRECISION_F64: render<double, false, false>(A); return 0;
    case RT_PRECISION_F32: render<float, true, true>(A); return 0;
    case RT_PRECISION_F32_LITERAL: render<float, false, false>(A); return 0;
    case RT_PRECISION_MIXED: render<double, false, false, true>(A); return 0;
    case RT_PRECISION_MIXED_FIXED: render<double, false, false, true, true>(A); return 0;
    default: return -1;
    }
}

/* RT_PRECISION_MIXED_FIXED with the accumulator words themselves: words[(index * 3 + channel) * 2 + {0, 1}] = the sums of
 * the samples' a and b words (units of 2^-32 and of 2^-64), what the GPU's accumulators hold up to carries between the two. */
int32_t oracle_render_fixed_words(uint64_t *words, int64_t width, int64_t height, rt_rect region, const rt_camera_render_data *cam,
                                  const rt_sphere *spheres, int64_t n_spheres, const rt_material *materials, int64_t n_materials,
                                  uint64_t seed, int64_t sample_begin, int64_t sample_count, int32_t threads) {
    render_args A{width, height, region, cam, spheres, n_spheres, materials, n_materials, seed,
                  sample_begin, sample_count, 1, nullptr, nullptr, nullptr, nullptr, threads};
    A.words = words;
    render<double, false, false, true, true>(A);
    return 0;
}

/* The mixed mode's filter for one ray and one sphere, with its intermediate values:
 * out = {lowered discriminant per unit direction w - thr, hs^2 - (w - thr) (the filter's stand-in for c of main.odin:248),
 *        hs (the center's parameter relative to the origin), behind, in_range (0/1), kept (0/1),
 *        bound on |c| for the next segment, |d| as the filter computes it}.
 * start_cc_bound >= 0: the sphere is the one the segment starts on, with that bound on its |c| (main.odin:248). */
void oracle_mixed_filter(const double *o, const double *d, const double *c, double radius, double reach, double start_cc_bound, double *out) {
    ray_t<double> r{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}};
    const filter_ray f = make_filter_ray(r, static_cast<float>(reach));
    const vec3<float> cf{(float)c[0], (float)c[1], (float)c[2]};
    const float Kn = sphere_constant_lowered(cf.x, cf.y, cf.z, (float)radius);
    float h, w;
    mixed_filter_values(f, cf, Kn, h, w);
    const float hs = h - f.s_o, dl = w - f.thr;
    out[0] = dl; out[1] = std::fmaf(hs, hs, -dl); out[2] = hs; out[3] = f.behind; out[4] = f.in_range ? 1.0 : 0.0;
    const bool on_start = start_cc_bound >= 0.0 && f.in_range && static_cast<float>(start_cc_bound) < 9.0e-4f * (f.nd * f.behind);
    out[5] = mixed_filter(f, cf, Kn, on_start) ? 1.0 : 0.0;
    out[6] = 7.2e-15f * f.reach_o * f.reach_o;
    out[7] = f.nd;
}

/* ray_sphere_intersection with explicit inputs (o, d, c are 3 doubles each) */
double oracle_ray_sphere_intersection(int32_t precision, const double *o, const double *d, const double *c,
                                      double radius, double t_min, double t_max) {
    if (precision == RT_PRECISION_F64) {
        ray_t<double> r{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}};
        return ray_sphere_intersection<double>(r, {c[0], c[1], c[2]}, radius, t_min, t_max);
    }
    ray_t<float> r{{(float)o[0], (float)o[1], (float)o[2]}, {(float)d[0], (float)d[1], (float)d[2]}};
    vec3<float> cf{(float)c[0], (float)c[1], (float)c[2]};
    if (precision == RT_PRECISION_F32) {
        /* the filter's reach for a world of this one sphere */
        const double far = std::sqrt((double)cf.x * cf.x + (double)cf.y * cf.y + (double)cf.z * cf.z) + std::fabs((double)(float)radius);
        const bool ok = std::fabs((double)(float)radius) >= 1.0 / 1048576.0 && far <= 1048576.0;
        float reach = static_cast<float>(far);
        if (static_cast<double>(reach) < far) reach = std::nextafterf(reach, INFINITY);
        const filter_ray f = make_filter_ray(ray_t<double>{{r.origin.x, r.origin.y, r.origin.z}, {r.direction.x, r.direction.y, r.direction.z}}, ok ? reach : FLT_MAX);
        return ray_sphere_intersection_expanded(expand_ray(r), f, cf, sphere_constant_lowered(cf.x, cf.y, cf.z, (float)radius), (float)radius, (float)t_min, (float)t_max);
    }
    return ray_sphere_intersection<float>(r, cf, (float)radius, (float)t_min, (float)t_max);
}

/* material_scatter for one hit; the draws are those of (seed, pixel, sample, bounce).
 * out = {origin xyz, direction xyz, attenuation xyz}; returns ok (0/1). */
int32_t oracle_material_scatter(int32_t precision, const rt_material *m, const double *ray_o, const double *ray_d,
                                const double *hit_p, const double *hit_normal, int32_t front_face,
                                uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, double *out) {
    draw_stream g(keys_from_seed(seed), pixel, sample, bounce);
    if (precision == RT_PRECISION_F64) {
        material_t<double> mm{m->type, cvt<double>(m->albedo), m->param1};
        ray_t<double> r{{ray_o[0], ray_o[1], ray_o[2]}, {ray_d[0], ray_d[1], ray_d[2]}};
        hit_record<double> h{{hit_p[0], hit_p[1], hit_p[2]}, {hit_normal[0], hit_normal[1], hit_normal[2]}, front_face != 0};
        auto s = material_scatter(mm, r, h, g);
        double tmp[9] = {s.ray_out.origin.x, s.ray_out.origin.y, s.ray_out.origin.z, s.ray_out.direction.x, s.ray_out.direction.y,
                         s.ray_out.direction.z, s.attenuation.x, s.attenuation.y, s.attenuation.z};
        std::memcpy(out, tmp, sizeof tmp);
        return s.ok;
    }
    material_t<float> mm{m->type, cvt<float>(m->albedo), (float)m->param1};
    ray_t<float> r{{(float)ray_o[0], (float)ray_o[1], (float)ray_o[2]}, {(float)ray_d[0], (float)ray_d[1], (float)ray_d[2]}};
    hit_record<float> h{{(float)hit_p[0], (float)hit_p[1], (float)hit_p[2]},
                        {(float)hit_normal[0], (float)hit_normal[1], (float)hit_normal[2]}, front_face != 0};
    auto s = precision == RT_PRECISION_F32 ? material_scatter<float, true>(mm, r, h, g) : material_scatter<float, false>(mm, r, h, g);
    double tmp[9] = {s.ray_out.origin.x, s.ray_out.origin.y, s.ray_out.origin.z, s.ray_out.direction.x, s.ray_out.direction.y,
                     s.ray_out.direction.z, s.attenuation.x, s.attenuation.y, s.attenuation.z};
    std::memcpy(out, tmp, sizeof tmp);
    return s.ok;
}

/* ---- host-side procedures around the path (callers and consumers) ---------------- */

/* lookat  main.odin:65-71.  out = right, up, forward (9 doubles) */
void oracle_lookat(const double *position, const double *target, const double *axis_up, double *out) {
    vec3<double> p{position[0], position[1], position[2]}, t{target[0], target[1], target[2]}, a{axis_up[0], axis_up[1], axis_up[2]};
    vec3<double> forward = normalize(t - p);
    vec3<double> right = normalize(cross(forward, a));
    vec3<double> up = cross(right, forward);
    double tmp[9] = {right.x, right.y, right.z, up.x, up.y, up.z, forward.x, forward.y, forward.z};
    std::memcpy(out, tmp, sizeof tmp);
}

/* calculate_camera_render_data  main.odin:450-485 */
void oracle_calculate_camera_render_data(const rt_camera_settings *cs, rt_camera_render_data *out) {
    const double TAU = 6.28318530717958647692528676655900576; /* math.TAU */
    auto turns_to_radians = [&](double t) { return t * TAU; };   /* main.odin:23 */
    vec3<double> position = cvt<double>(cs->position), right = cvt<double>(cs->right), up = cvt<double>(cs->up), forward = cvt<double>(cs->forward);
    double half_y = cs->focus_distance * std::tan(turns_to_radians(cs->vfov * 0.5));
    double half_x = half_y * cs->aspect_ratio;
    /* view_plane_half_size * 2 / image_size * {1, -1}, element-wise, left to right */
    double pdx = half_x * 2 / cs->image_size_x * 1.0;
    double pdy = half_y * 2 / cs->image_size_y * -1.0;
    vec3<double> du = right * pdx, dv = up * pdy;
    double tlx = half_x * -1.0 + pdx * 0.5;
    double tly = half_y * 1.0 + pdy * 0.5;
    vec3<double> tl = position + forward * cs->focus_distance + right * tlx + up * tly;
    double dof_radius = cs->focus_distance * std::tan(turns_to_radians(cs->depth_of_field_angle * 0.5));
    vec3<double> dof_u = right * dof_radius, dof_v = up * dof_radius;
    auto put = [](rt_v3 &d, vec3<double> s) { d.x = s.x; d.y = s.y; d.z = s.z; };
    put(out->position, position); put(out->pixel_delta_u, du); put(out->pixel_delta_v, dv);
    put(out->view_plane_top_left_pixel_center, tl); put(out->dof_disk_u, dof_u); put(out->dof_disk_v, dof_v);
    out->depth_of_field_angle = cs->depth_of_field_angle;
    out->samples_per_pixel = cs->samples_per_pixel;
    out->max_ray_bounces = cs->max_ray_bounces;
}

/* the quantiser of serialize_ppm  main.odin:812-814: u8(256 * min(x, 0.999)) */
void oracle_quantize_rgb8(const double *image, int64_t n_values, uint8_t *out) {
    for (int64_t i = 0; i < n_values; ++i) {
        double q = 256 * std::min(image[i], 0.999);
        out[i] = static_cast<uint8_t>(static_cast<int64_t>(q)); /* Odin's u8() cast truncates and wraps */
    }
}

/* serialize_ppm  main.odin:791-824.  returns bytes written (dst must hold 32 + 12*w*h). */
int64_t oracle_serialize_ppm(const double *image, int64_t width, int64_t height, char *dst) {
    int64_t pos = 0;
    {   /* fmt.sbprintfln("P3\n%v %v\n255") */
        char hdr[64];
        int n = std::snprintf(hdr, sizeof hdr, "P3\n%lld %lld\n255\n", (long long)width, (long long)height);
        std::memcpy(dst, hdr, n);
        pos = n;
    }
    auto channel = [](char *d, uint8_t u, char sep) { /* serialize_channel  main.odin:797-806 */
        d[0] = static_cast<char>('0' + ((u / 100) % 10));
        d[1] = static_cast<char>('0' + ((u / 10) % 10));
        d[2] = static_cast<char>('0' + (u % 10));
        d[3] = sep;
        d[0] = d[0] == '0' ? ' ' : d[0];
        d[1] = (d[0] == ' ' && d[1] == '0') ? ' ' : d[1];
    };
    for (int64_t i = 0; i < width * height; ++i) {
        uint8_t q[3];
        oracle_quantize_rgb8(image + i * 3, 3, q);
        channel(dst + pos + 0, q[0], ' ');
        channel(dst + pos + 4, q[1], ' ');
        channel(dst + pos + 8, q[2], '\n');
        pos += 12;
    }
    return pos;
}

} // extern "C"
