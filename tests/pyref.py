"""Second, independent restatement of the reference's hot path in plain Python floats (IEEE f64).

Written from ray-tracer/main.odin directly, not from oracle/rt_oracle.cpp, so that a
transcription slip in either shows up as a bitwise difference between the two.  Python's
float arithmetic and math.sqrt are IEEE-754 double with correct rounding, the same as the
reference's f64, so agreement with the C++ oracle is expected bit for bit.
Slow: use for images of a few hundred pixel-samples only.
"""
import math
from fractions import Fraction

M32 = 0xFFFFFFFF
F64_EPSILON = 2.220446049250313e-16
F64_MAX = 1.7976931348623157e308
REJECTION_CAP = 64


# ---- counter RNG (the build's own; replaces core:math/rand at the reference's call sites)
def mix32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x21F0AAAD) & M32
    x ^= x >> 15
    x = (x * 0x735A2D97) & M32
    x ^= x >> 15
    return x


def seed_keys(seed):
    k0 = mix32((seed & M32) ^ 0x9E3779B9)
    k1 = mix32(((seed >> 32) & M32) ^ k0 ^ 0x85EBCA6B)
    return k0, k1


def rng_u32(seed, pixel, sample, bounce, draw):
    k0, k1 = seed_keys(seed)
    kp = mix32(k0 ^ pixel)
    wp = mix32((k1 + pixel) & M32)
    ks = mix32((kp + 0x9E3779B9 * (sample + 1)) & M32)
    kb = mix32(ks ^ ((0x85EBCA6B * (bounce + 1)) & M32))
    return mix32((kb + 0xC2B2AE35 * (draw + 1)) & M32) ^ wp


class Gen:
    """the draws of one (pixel, sample, bounce), in order"""

    def __init__(self, seed, pixel, sample, bounce):
        self.args = (seed, pixel, sample, bounce)
        self.n = 0

    def float64(self):  # rand.float64()
        u = rng_u32(*self.args, self.n)
        self.n += 1
        return (u >> 8) * (1.0 / 16777216.0)

    def float64_range(self, lo, hi):  # rand.float64_range()
        val = (hi - lo) * self.float64() + lo
        if val >= hi:
            val = max(lo, hi * (1 - F64_EPSILON))
        return val


# ---- linalg (main.odin:28-57)
def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def vadd(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def vsub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def vmul(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def smul(s, a):
    return (s * a[0], s * a[1], s * a[2])


def vdiv(a, s):
    return (a[0] / s, a[1] / s, a[2] / s)


def neg(a):
    return (-a[0], -a[1], -a[2])


def normalize(v):
    return vdiv(v, math.sqrt(dot(v, v)))


def is_near_zero(v):
    e = 1e-8
    return abs(v[0]) <= e and abs(v[1]) <= e and abs(v[2]) <= e


# ---- optics (main.odin:77-80, 111-122, 168-173)
def reflect(v, n):
    return vsub(v, smul(2 * dot(v, n), n))


def reflectance_at_normal_incidence(rel):
    r0 = (1.0 - rel) / (1.0 + rel)
    r0 *= r0
    return r0


def reflectance_schlick_approximation(cos_i, r0):
    a = 1.0 - cos_i
    return r0 + (1.0 - r0) * a * a * a * a * a


def refract(v, n, rel):
    cos_theta = min(dot(neg(v), n), 1.0)
    perp = smul(rel, vadd(v, smul(cos_theta, n)))
    par = smul(-math.sqrt(max(1.0 - dot(perp, perp), 0.0)), n)
    return vadd(perp, par)


# ---- random (main.odin:190-223)
def random_unit_vector(gen):
    p = (0.0, 0.0, 0.0)
    lsq = 0.0
    for _ in range(REJECTION_CAP):
        p = (gen.float64_range(-1, 1 + F64_EPSILON), gen.float64_range(-1, 1 + F64_EPSILON), gen.float64_range(-1, 1 + F64_EPSILON))
        lsq = dot(p, p)
        if 0 < lsq <= 1:
            break
    return vdiv(p, math.sqrt(lsq))


def random_point_on_disk(gen):
    p = (0.0, 0.0, 0.0)
    for _ in range(REJECTION_CAP):
        p = (gen.float64_range(-1, 1 + F64_EPSILON), gen.float64_range(-1, 1 + F64_EPSILON), 0.0)
        if dot(p, p) <= 1:
            break
    return p


# ---- raycast (main.odin:240-263)
def ray_sphere_intersection(ro, rd, center, radius, t_min, t_max):
    t = t_max
    o_c = vsub(center, ro)
    a = dot(rd, rd)
    h = dot(rd, o_c)
    c = dot(o_c, o_c) - radius * radius
    discriminant = h * h - a * c
    if discriminant < 0:
        return t
    sqrtd = math.sqrt(discriminant)
    root = (h - sqrtd) / a
    if root < t_min or t_max <= root:
        root = (h + sqrtd) / a
        if root < t_min or t_max <= root:
            return t
    return root


def reflectance_fresnel(cos_i, sin_i, n1, n2):  # main.odin:84-109
    subexp = (n1 / n2) * sin_i
    subexp = 1.0 - subexp * subexp
    subexp = math.sqrt(subexp)
    x1, x2 = n1 * cos_i, n2 * subexp
    r_s = (x1 - x2)
    r_s /= (x1 + x2)
    r_s *= r_s
    x1, x2 = n1 * subexp, n2 * cos_i
    r_p = (x1 - x2)
    r_p /= (x1 + x2)
    r_p *= r_p
    return (r_s + r_p) * 0.5


# ---- materials (main.odin:292-410); material = (type, albedo, param1)
def material_scatter(material, ro, rd, hit_p, hit_n, front_face, gen):
    mtype, albedo, param1 = material
    if mtype == 0:
        out = vadd(hit_n, random_unit_vector(gen))
        if is_near_zero(out):
            out = hit_n
        return (hit_p, out), albedo, True
    if mtype in (1, 3, 4, 5, 6):
        out = normalize(reflect(rd, hit_n))
        out = vadd(out, smul(param1, random_unit_vector(gen)))
        attenuation = albedo
        if mtype >= 3:  # USE_METAL_FRESNEL (main.odin:324-347): type 3 = METAL_FRESNEL_KIND 2, 4 = kind 0, 5 = kind 1 as written, 6 = kind 1 as the papers have it
            cos_theta = min(dot(neg(normalize(rd)), hit_n), 1.0)
            if mtype == 3:
                sin_theta = math.sqrt(1.0 - cos_theta * cos_theta)
                t = reflectance_fresnel(cos_theta, sin_theta, 1.0, 1.27035)
            else:
                rel = float(Fraction(100000, 127035))      # 1.0/1.27035 as a constant expression: exact quotient, rounded once
                r0 = (1.0 - rel) / (1.0 + rel)
                r0 *= r0
                x = 1.0 - cos_theta
                t = r0 + (1.0 - r0) * x * x * x * x * x    # reflectance_schlick_approximation(cos_theta, r0), main.odin:119-122
                if mtype == 5:                              # Hoffman as main.odin:132-144 reads: schlick(r0, cos_i), integer quotients 17 and 8
                    a1 = 1.0 - r0
                    t = cos_theta + (1.0 - cos_theta) * a1 * a1 * a1 * a1 * a1
                    a = 17.0 * (r0 - 0.5) + 8.0 * (1.0 - r0)
                if mtype == 6:                              # Hoffman, arguments and constants as the papers have them
                    a = (823543.0 / 46656.0) * (r0 - 0.5) + (49.0 / 6.0) * (1.0 - r0)
                if mtype >= 5:
                    x2 = x * x
                    x3 = x2 * x
                    t = t - a * cos_theta * (x3 * x3)
            attenuation = vadd(smul(1.0 - t, albedo), smul(t, (1.0, 1.0, 1.0)))  # math.lerp(albedo, 1, t)
        return (hit_p, out), attenuation, dot(out, hit_n) > 0
    ri = param1
    src = 1.0 if ri >= 1.0 else 1.0 / ri
    dst = ri if ri >= 1.0 else 1.0
    if not front_face:
        src, dst = dst, src
    rel = src / dst
    unit = normalize(rd)
    cos_theta = min(dot(neg(unit), hit_n), 1.0)
    sin2 = 1.0 - cos_theta * cos_theta
    must_reflect = (rel * rel * sin2) > 1.0
    if must_reflect or reflectance_schlick_approximation(cos_theta, reflectance_at_normal_incidence(rel)) > gen.float64():
        out = reflect(unit, hit_n)
    else:
        out = refract(unit, hit_n, rel)
    return (hit_p, out), (1.0, 1.0, 1.0), True


def background_color(rd):
    a, b = (1.0, 1.0, 1.0), (0.5, 0.7, 1.0)
    t = 0.5 * (rd[1] + 1.0)
    return vadd(smul(1 - t, a), smul(t, b))  # math.lerp: a*(1-t) + b*t (commutes bitwise per element)


# ---- render_region (main.odin:510-591)
def render_region(width, height, region, cam, spheres, materials, seed, want_linear=False):
    """cam: dict of the camera_render_data fields; spheres: [(center, radius, material_index)];
    returns {(u, v): (r, g, b)} gamma-space (or linear means) and the segment count."""
    out = {}
    segments = 0
    spp = cam["samples_per_pixel"]
    contribution = 1.0 / float(spp)
    for v in range(region[1], region[3]):
        for u in range(region[0], region[2]):
            pixel_color = (0.0, 0.0, 0.0)
            pixel = u + v * width
            for s in range(spp):
                g = Gen(seed, pixel, s, 0)
                ox = g.float64_range(-0.5, 0.5)
                oy = g.float64_range(-0.5, 0.5)
                psp = vadd(vadd(cam["view_plane_top_left_pixel_center"], smul(float(u) + ox, cam["pixel_delta_u"])),
                           smul(float(v) + oy, cam["pixel_delta_v"]))
                ro = cam["position"]
                if cam["depth_of_field_angle"] > 0.0:
                    d = random_point_on_disk(g)
                    ro = vadd(ro, vadd(smul(d[0], cam["dof_disk_u"]), smul(d[1], cam["dof_disk_v"])))
                rd = vsub(psp, ro)
                sample_color = (1.0, 1.0, 1.0)
                for bounce in range(cam["max_ray_bounces"] + 1):
                    segments += 1
                    closest_t = F64_MAX
                    hit_index = -1
                    for i, (c, r, _) in enumerate(spheres):
                        t = ray_sphere_intersection(ro, rd, c, r, 0.001, closest_t)
                        if t < closest_t:
                            closest_t, hit_index = t, i
                    if hit_index >= 0:
                        c, r, mi = spheres[hit_index]
                        p = vadd(ro, smul(closest_t, rd))
                        n = vdiv(vsub(p, c), r)
                        front = dot(rd, n) < 0.0
                        n = n if front else neg(n)
                        (nro, nrd), att, ok = material_scatter(materials[mi], ro, rd, p, n, front, Gen(seed, pixel, s, bounce + 1))
                        if ok:
                            ro, rd = nro, nrd
                            sample_color = vmul(sample_color, att)
                        else:
                            sample_color = (0.0, 0.0, 0.0)
                            break
                    else:
                        rd = normalize(rd)
                        sample_color = vmul(sample_color, background_color(rd))
                        break
                pixel_color = vadd(pixel_color, sample_color)
            pixel_color = smul(contribution, pixel_color)
            if not want_linear:
                pixel_color = (math.sqrt(pixel_color[0]), math.sqrt(pixel_color[1]), math.sqrt(pixel_color[2]))
            out[(u, v)] = pixel_color
    return out, segments
