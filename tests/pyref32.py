"""Independent statement of the GPU throughput arithmetic (precision RT_PRECISION_F32) in Python.

Every f32 operation is a numpy float32 scalar operation (IEEE binary32, round to nearest even),
every fused multiply-add is evaluated exactly in rational arithmetic and rounded once, and the
sample colours are summed as Q32.32 integers.  Written from the description of the arithmetic
(the reference's expressions of ray-tracer/main.odin in f32, with the ray/sphere test in two
stages -- a conservative filter, then the roots --, the start-sphere rule and reciprocal scaling), not from the C++ oracle, so that the
oracle's f32 mode and the GPU kernel are pinned by something other than each other.
Very slow (a rational fma per sphere test): a few thousand samples at most.
"""
import math
from fractions import Fraction

import numpy as np

import pyref

F = np.float32
ZERO, ONE, TWO, HALF = F(0.0), F(1.0), F(2.0), F(0.5)
F32_MAX = np.finfo(np.float32).max
F32_EPS = np.finfo(np.float32).eps
T_MIN = F(0.001)  # SHADOW_ACNE_RAY_OFFSET (main.odin:538)
REJECTION_CAP = 64


def fma(a, b, c):
    """round_to_f32(a*b + c), one rounding"""
    a, b, c = F(a), F(b), F(c)
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        with np.errstate(all="ignore"):
            return F(np.float64(a) * np.float64(b) + np.float64(c))  # inf/nan propagate the same way
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    if exact == 0:
        # the sign of an exact zero sum: +0 unless both addends are -0 (round to nearest)
        prod_negative = (np.signbit(a) != np.signbit(b))
        return F(-0.0) if (prod_negative and np.signbit(c)) else F(0.0)
    with np.errstate(over="ignore"):
        guess = F(float(exact))  # double rounding possible: repair below
    if not np.isfinite(guess):
        return guess
    best = guess
    for cand in (np.nextafter(guess, F(-np.inf)), np.nextafter(guess, F(np.inf))):
        if not np.isfinite(cand):
            continue
        d_best, d_cand = abs(exact - Fraction(float(best))), abs(exact - Fraction(float(cand)))
        if d_cand < d_best or (d_cand == d_best and (int(np.float32(cand).view(np.uint32)) & 1) == 0):
            best = cand
    return F(best)


# ---- vectors of float32
def v3(x, y, z):
    return (F(x), F(y), F(z))


def vadd(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def vsub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def vneg(a):
    return (-a[0], -a[1], -a[2])


def vmul(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def smul(s, a):
    return (s * a[0], s * a[1], s * a[2])


def dot(a, b):  # main.odin:28-30, left to right
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def scale_down(v, s):
    """a vector divided by a scalar: one reciprocal, three products"""
    r = ONE / s
    return (v[0] * r, v[1] * r, v[2] * r)


def normalize(v):
    return scale_down(v, np.sqrt(dot(v, v)))


def reflect(v, n):  # main.odin:77-80
    return vsub(v, smul(TWO * dot(v, n), n))


def refract(v, n, rel):  # main.odin:168-173
    cos_theta = min(dot(vneg(v), n), ONE)
    perp = smul(rel, vadd(v, smul(cos_theta, n)))
    par = smul(-np.sqrt(max(ONE - dot(perp, perp), ZERO)), n)
    return vadd(perp, par)


# ---- draws in f32
class Gen32:
    def __init__(self, seed, pixel, sample, bounce):
        self.args = (seed, pixel, sample, bounce)
        self.n = 0

    def uniform(self):
        u = pyref.rng_u32(*self.args, self.n)
        self.n += 1
        return F(u >> 8) * F(1.0 / 16777216.0)

    def range(self, lo, hi):
        val = (hi - lo) * self.uniform() + lo
        if val >= hi:
            val = max(lo, hi * (ONE - F32_EPS))
        return val


def random_unit_vector(g):  # main.odin:194-203; 1 + F64_EPSILON rounds to 1 in f32
    hi = F(1.0 + 2.220446049250313e-16)
    p, lsq = (ZERO, ZERO, ZERO), ZERO
    for _ in range(REJECTION_CAP):
        p = (g.range(-ONE, hi), g.range(-ONE, hi), g.range(-ONE, hi))
        lsq = dot(p, p)
        if ZERO < lsq <= ONE:
            break
    return scale_down(p, np.sqrt(lsq))


def random_point_on_disk(g):  # main.odin:214-223
    hi = F(1.0 + 2.220446049250313e-16)
    p = (ZERO, ZERO, ZERO)
    for _ in range(REJECTION_CAP):
        p = (g.range(-ONE, hi), g.range(-ONE, hi), ZERO)
        if dot(p, p) <= ONE:
            break
    return p


# ---- the two-stage ray/sphere test
def filter_constant(c, r):
    """-K' of the filter: K = |c|^2 - r^2 of the f32 centre and radius lowered by 2^-16 (|c|^2 + r^2), formed in f64;
    K' is rounded down to f32, i.e. -K' up"""
    c2 = float(c[0]) * float(c[0]) + float(c[1]) * float(c[1]) + float(c[2]) * float(c[2])
    r2 = float(r) * float(r)
    lowered = (c2 - r2) - (c2 + r2) * (1.0 / 65536.0)
    kn = F(-lowered)
    if float(kn) < -lowered:
        kn = np.nextafter(kn, F(np.inf))
    return F(kn)


def filter_reach(spheres):
    """max |c| + |r| over the f32 spheres, rounded up to f32; FLT_MAX when a sphere is outside the range the
    filter's error bound needs (then every ray keeps every sphere)"""
    reach, ok = 0.0, True
    for c, r, _ in spheres:
        far = math.sqrt(float(c[0]) * float(c[0]) + float(c[1]) * float(c[1]) + float(c[2]) * float(c[2])) + abs(float(r))
        reach = max(reach, far)
        if not (abs(float(r)) >= 1.0 / 1048576.0) or not (far <= 1048576.0):
            ok = False
    if not ok:
        return F32_MAX
    out = F(reach)
    if float(out) < reach:
        out = np.nextafter(out, F(np.inf))
    return F(out)


class Segment:
    """per-segment quantities of the test"""

    def __init__(self, ro, rd, reach):
        self.o, self.d = ro, rd
        o_o = dot(ro, ro)
        self.a = dot(rd, rd)
        with np.errstate(divide="ignore"):
            self.inv_a = ONE / self.a
        # stage 1, the conservative filter: unit direction e, foot point f of the coordinate origin on the line
        nd = np.sqrt(self.a)
        with np.errstate(divide="ignore"):
            inv_nd = ONE / nd
        self.e = (rd[0] * inv_nd, rd[1] * inv_nd, rd[2] * inv_nd)
        self.s_o = dot(ro, self.e)
        ft = (fma(-self.s_o, self.e[0], ro[0]), fma(-self.s_o, self.e[1], ro[1]), fma(-self.s_o, self.e[2], ro[2]))
        self.thr = fma(F(-1.0 / 65536.0), o_o, dot(ft, ft))
        self.f2 = (ft[0] + ft[0], ft[1] + ft[1], ft[2] + ft[2])
        reach_o = reach + (abs(ro[0]) + abs(ro[1]) + abs(ro[2]))
        self.behind = F(1.0 / 524288.0) * reach_o
        self.in_range = bool(self.a >= F(2.0 ** -40) and self.a <= F(2.0 ** 40) and o_o <= F(2.0 ** 40) and reach <= F(1048576.0))

    def filter_values(self, c, kn):
        """(hs, w - thr): the centre's parameter relative to the origin and the lowered discriminant, per unit direction"""
        h = fma(c[2], self.e[2], fma(c[1], self.e[1], c[0] * self.e[0]))
        w = fma(c[2], self.f2[2], fma(c[1], self.f2[1], fma(c[0], self.f2[0], fma(h, h, kn))))
        return h - self.s_o, w - self.thr, w

    def keeps(self, c, kn):
        """False only when the sphere's exact discriminant is certainly negative, or the line certainly meets the sphere
        only behind an origin certainly outside it"""
        if not self.in_range:
            return True
        hs, dl, w = self.filter_values(c, kn)
        if w < self.thr:
            return False
        if hs < -self.behind and fma(hs, hs, -dl) > ZERO:
            return False
        return True

    def q_of(self, c):
        oc = vsub(c, self.o)
        q = fma(oc[2], self.d[2], fma(oc[1], self.d[1], oc[0] * self.d[0])) * self.inv_a
        return oc, q

    def from_surface(self, c, t_max):
        """the sphere the segment starts on: c = 0, roots 0 and 2q"""
        _, q = self.q_of(c)
        root = q + q
        if q > ZERO and root >= T_MIN and root < t_max:
            return root
        return t_max

    def test(self, c, kn, radius, t_max):
        d = self.d
        if not self.keeps(c, kn):
            return t_max
        oc, q = self.q_of(c)
        l = (fma(-q, d[0], oc[0]), fma(-q, d[1], oc[1]), fma(-q, d[2], oc[2]))
        D = radius * radius - fma(l[2], l[2], fma(l[1], l[1], l[0] * l[0]))
        if not (D >= ZERO):
            return t_max
        s = np.sqrt(D * self.inv_a)
        root = q - s
        if root < T_MIN or t_max <= root:
            root = q + s
            if root < T_MIN or t_max <= root:
                return t_max
        return root


# ---- materials in f32 (main.odin:292-410); material = (type, albedo, param1)
def material_scatter(material, rd, hit_p, hit_n, front_face, g):
    mtype, albedo, param1 = material
    if mtype == 0:
        out = vadd(hit_n, random_unit_vector(g))
        if all(abs(x) <= F(1e-8) for x in out):
            out = hit_n
        return (hit_p, out), albedo, True
    if mtype in (1, 3, 4, 5, 6):
        out = normalize(reflect(rd, hit_n))
        out = vadd(out, smul(param1, random_unit_vector(g)))
        attenuation = albedo
        if mtype >= 4:  # METAL_FRESNEL_KIND 0 (Schlick) and 1 (Hoffman, as the papers define it): main.odin:333-339, :111-144
            cos_theta = min(dot(vneg(normalize(rd)), hit_n), ONE)
            rel = F(0.7871846341559413)
            r0 = (ONE - rel) / (ONE + rel)
            r0 = r0 * r0
            x = ONE - cos_theta
            t = r0 + (ONE - r0) * x * x * x * x * x
            if mtype == 5:   # as main.odin:133, :142 read: schlick(r0, cos_i), integer quotients 17 and 8
                a1 = ONE - r0
                t = cos_theta + (ONE - cos_theta) * a1 * a1 * a1 * a1 * a1
                a = F(17.0) * (r0 - HALF) + F(8.0) * (ONE - r0)
            if mtype == 6:   # as the cited papers define it
                a = F(823543.0 / 46656.0) * (r0 - HALF) + F(49.0 / 6.0) * (ONE - r0)
            if mtype >= 5:
                x2 = x * x
                x3 = x2 * x
                t = t - a * cos_theta * (x3 * x3)
            attenuation = vadd(smul(ONE - t, albedo), smul(t, (ONE, ONE, ONE)))
        elif mtype == 3:  # main.odin:326-328, :341-347, :84-109
            cos_theta = min(dot(vneg(normalize(rd)), hit_n), ONE)
            sin_theta = np.sqrt(ONE - cos_theta * cos_theta)
            n1, n2 = ONE, F(1.27035)
            sub = (n1 / n2) * sin_theta
            sub = np.sqrt(ONE - sub * sub)
            x1, x2 = n1 * cos_theta, n2 * sub
            r_s = (x1 - x2) / (x1 + x2)
            r_s = r_s * r_s
            x1, x2 = n1 * sub, n2 * cos_theta
            r_p = (x1 - x2) / (x1 + x2)
            r_p = r_p * r_p
            t = (r_s + r_p) * HALF
            attenuation = vadd(smul(ONE - t, albedo), smul(t, (ONE, ONE, ONE)))
        return (hit_p, out), attenuation, dot(out, hit_n) > ZERO
    ri = param1
    src = ONE if ri >= ONE else ONE / ri
    dst = ri if ri >= ONE else ONE
    if not front_face:
        src, dst = dst, src
    rel = src / dst
    unit = normalize(rd)
    cos_theta = min(dot(vneg(unit), hit_n), ONE)
    sin2 = ONE - cos_theta * cos_theta
    must_reflect = (rel * rel * sin2) > ONE
    reflects = must_reflect
    if not must_reflect:
        r0 = (ONE - rel) / (ONE + rel)
        r0 = r0 * r0
        a = ONE - cos_theta
        reflects = (r0 + (ONE - r0) * a * a * a * a * a) > g.uniform()
    out = reflect(unit, hit_n) if reflects else refract(unit, hit_n, rel)
    return (hit_p, out), (ONE, ONE, ONE), True


def to_q32_32(x):
    x = F(x)
    if not (x > ZERO):
        return 0
    if x >= F(4294967296.0):
        return 0xFFFFFFFF00000000
    hi = int(x)
    fr = x - F(hi)
    return (hi << 32) | int(fr * F(4294967296.0))


def render_region(width, height, region, cam, spheres, materials, seed):
    """cam: dict of the camera_render_data fields (f64 values); spheres: [(centre, radius, material_index)];
    materials: [(type, albedo, param1)].  Returns {(u, v): (gamma rgb as f64)}, {(u, v): (Q32.32 sums)} and the
    segment count."""
    cam32 = {k: (v3(*v) if isinstance(v, tuple) else v) for k, v in cam.items()}
    dof_angle = F(cam["depth_of_field_angle"])
    sph = [(v3(*c), F(r), mi) for c, r, mi in spheres]
    consts = [(filter_constant(c, r), ONE / r) for c, r, _ in sph]
    reach = filter_reach(sph)
    mats = [(t, v3(*alb), F(p1)) for t, alb, p1 in materials]
    spp = cam["samples_per_pixel"]
    image, sums, segments = {}, {}, 0
    with np.errstate(over="ignore", invalid="ignore"):
        for v in range(region[1], region[3]):
            for u in range(region[0], region[2]):
                q = [0, 0, 0]
                pixel = u + v * width
                for s in range(spp):
                    g = Gen32(seed, pixel, s, 0)
                    ox = g.range(F(-0.5), HALF)
                    oy = g.range(F(-0.5), HALF)
                    psp = vadd(vadd(cam32["view_plane_top_left_pixel_center"], smul(F(u) + ox, cam32["pixel_delta_u"])),
                               smul(F(v) + oy, cam32["pixel_delta_v"]))
                    ro = cam32["position"]
                    if dof_angle > ZERO:
                        d = random_point_on_disk(g)
                        ro = vadd(ro, vadd(smul(d[0], cam32["dof_disk_u"]), smul(d[1], cam32["dof_disk_v"])))
                    rd = vsub(psp, ro)
                    color = (ONE, ONE, ONE)
                    start = -1
                    for bounce in range(cam["max_ray_bounces"] + 1):
                        segments += 1
                        closest, hit = F32_MAX, -1
                        seg = Segment(ro, rd, reach)
                        if start >= 0:
                            t = seg.from_surface(sph[start][0], closest)
                            if t < closest:
                                closest, hit = t, start
                        for i, (c, r, _) in enumerate(sph):
                            if i == start:
                                continue
                            t = seg.test(c, consts[i][0], r, closest)
                            if t < closest:
                                closest, hit = t, i
                        if hit < 0:
                            rd = normalize(rd)
                            t = HALF * (rd[1] + ONE)
                            sky = vadd(smul(ONE - t, (ONE, ONE, ONE)), smul(t, v3(0.5, 0.7, 1.0)))  # math.lerp (main.odin:417)
                            color = vmul(color, sky)
                            break
                        c, r, mi = sph[hit]
                        p = vadd(ro, smul(closest, rd))
                        n = smul(consts[hit][1], vsub(p, c))
                        front = dot(rd, n) < ZERO
                        n = n if front else vneg(n)
                        (nro, nrd), att, ok = material_scatter(mats[mi], rd, p, n, front, Gen32(seed, pixel, s, bounce + 1))
                        if not ok:
                            color = (ZERO, ZERO, ZERO)
                            break
                        start, ro, rd = hit, nro, nrd
                        color = vmul(color, att)
                    for k in range(3):
                        q[k] += to_q32_32(color[k])
                sums[(u, v)] = tuple(q)
                contribution = 1.0 / float(spp)
                image[(u, v)] = tuple(math.sqrt((float(qk) * (1.0 / 4294967296.0)) * contribution) for qk in q)
    return image, sums, segments
