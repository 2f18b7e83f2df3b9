"""The f32 candidate filter of RT_PRECISION_MIXED, on the CPU (oracle restatement of the kernel's filter).

The filter may keep too much, never too little: a sphere it drops must be one that the reference's
f64 test (main.odin:245-261) cannot report as hit.  Checked two ways: against exact rational
arithmetic on adversarial single cases, and by auditing every dropped sphere of whole renders.
"""
from fractions import Fraction

import numpy as np
import pytest

U = Fraction(1, 2 ** 24)


def exact_terms(o, d, c, r):
    """exact h, a*C and discriminant of main.odin:245-249 for f64 inputs (rational arithmetic)"""
    o, d, c = ([Fraction(float(x)) for x in v] for v in (o, d, c))
    r = Fraction(float(r))
    oc = [c[k] - o[k] for k in range(3)]
    a = sum(x * x for x in d)
    h = sum(d[k] * oc[k] for k in range(3))
    cc = sum(x * x for x in oc) - r * r
    return a, h, cc, h * h - a * cc


def norm_upper(v):
    return Fraction(float(np.sqrt(float(sum(Fraction(float(x)) ** 2 for x in v))))) * Fraction(1000001, 1000000)


def cases(rng, n):
    """rays and spheres chosen to stress the cancellation in the expanded discriminant: far cameras,
    grazing rays (the line passes the sphere at almost exactly its radius), tiny and huge spheres,
    short and long directions"""
    for _ in range(n):
        scale = 10.0 ** rng.uniform(-3, 4)
        c = rng.normal(size=3) * scale * rng.choice([0.0, 0.01, 1.0])
        r = scale * 10.0 ** rng.uniform(-3, 0.5) * rng.choice([1.0, -1.0], p=[0.9, 0.1])
        o = c + rng.normal(size=3) * abs(r) * 10.0 ** rng.uniform(-1, 3)
        kind = rng.integers(0, 3)
        if kind == 0:     # towards a point at distance ~|r| from the centre: grazing
            t = rng.normal(size=3)
            t /= np.linalg.norm(t)
            oc = c - o
            t -= oc * (t @ oc) / (oc @ oc)
            t /= np.linalg.norm(t)
            target = c + t * abs(r) * (1.0 + rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-9, -2))
            d = target - o
        elif kind == 1:   # through the sphere
            d = (c + rng.normal(size=3) * abs(r) * 0.5) - o
        else:             # anywhere
            d = rng.normal(size=3)
        d = d * 10.0 ** rng.uniform(-4, 4) / max(np.linalg.norm(d), 1e-300)
        yield o, d, c, r


def test_filter_never_drops_a_sphere_the_exact_discriminant_accepts(oracle):
    rng = np.random.default_rng(2024)
    worst_low, worst_high, worst_h, worst_rounding, kept, total = None, None, 0.0, Fraction(0), 0, 0
    for o, d, c, r in cases(rng, 6000):
        cf = np.float32(c).astype(np.float64)
        reach = float(np.linalg.norm(cf) + abs(np.float32(r))) * 1.000001
        f = oracle.mixed_filter(o, d, c, r, reach)
        a, h, cc, disc = exact_terms(o, d, c, r)
        if not f["in_range"]:
            assert f["kept"]
            continue
        total += 1
        kept += f["kept"]
        S = norm_upper(c) + norm_upper(o) + abs(Fraction(float(r)))
        unit = U * S * S
        # the filter works per unit direction: its lowered discriminant w - thr stands for disc / a = r^2 - dist(centre, line)^2
        lowered, exact = Fraction(f["disc"]), disc / a
        # never below the exact one, and by at least the 24 u S^2 the behind-the-origin rule relies on:
        # the margin (>= 85 u S^2) exceeds the rounding (<= 43 u S^2) by more than that
        assert lowered >= exact + 24 * unit, (o, d, c, r)
        gap = (lowered - exact) / unit
        # the rounding itself (what is left after taking the margin 2^-16 (|c|^2 + r^2 + |o|^2) of the f32 inputs off): within the derived 43 u S^2
        c32, o32, r32 = (Fraction(float(np.float32(x))) for x in c), (Fraction(float(np.float32(x))) for x in o), Fraction(float(np.float32(r)))
        margin = (sum(x * x for x in c32) + r32 * r32 + sum(x * x for x in o32)) / 65536
        rounding = (lowered - exact - margin) / unit
        assert abs(rounding) <= 43, (o, d, c, r)
        worst_rounding = max(worst_rounding, abs(rounding))
        worst_low = gap if worst_low is None else min(worst_low, gap)
        worst_high = gap if worst_high is None else max(worst_high, gap)
        if disc >= 0:
            assert f["disc"] >= 0
        # the centre's parameter relative to the origin, per unit direction: off by at most 12 u S
        h_unit = float(h) / float(np.sqrt(float(a)))
        err_h = abs(f["h"] - h_unit) / float(U * S)
        assert err_h <= 12.0, (o, d, c, r)
        worst_h = max(worst_h, err_h)
        # the behind-the-origin rule only fires when the exact signs agree
        if f["h"] < -f["behind"] and f["m"] > 0:
            assert h < 0 and cc > 0
        if disc >= 0 and not (h < 0 and cc > 0):
            assert f["kept"]
    print(f"lowered - exact discriminant (per unit direction) in units of u*S^2: min {float(worst_low):.2f}, max {float(worst_high):.2f}; "
          f"rounding alone at most {float(worst_rounding):.2f} (bound 43); error of the centre's parameter: at most {worst_h:.2f} u*S (bound 12); kept {kept} of {total}")
    assert worst_low > 24


def test_start_sphere_rule_only_drops_what_the_reference_misses(oracle):
    """A segment that starts on a sphere (its origin is the hit point the reference computed on it) and leaves it:
    the filter may drop that sphere only if main.odin:240-263 evaluated in f64 reports no hit.  Hit points come from
    the reference's own root formula, from near and far previous origins; outgoing directions include grazing ones
    (almost tangent, where the far root of the start sphere creeps up towards t_min) and very short ones."""
    rng = np.random.default_rng(77)
    dropped = kept = hits_kept = 0
    for _ in range(8000):
        scale = 10.0 ** rng.uniform(-2, 3)
        c = rng.normal(size=3) * scale * rng.choice([0.0, 1.0, 30.0])
        r = scale * 10.0 ** rng.uniform(-2, 0.5)
        # previous segment: from o0 towards a point inside the sphere
        o0 = c + rng.normal(size=3) * r * 10.0 ** rng.uniform(0.2, 4)
        d0 = (c + rng.normal(size=3) * r * 0.55) - o0
        d0 *= 10.0 ** rng.uniform(-2, 2) / np.linalg.norm(d0)
        t = oracle.ray_sphere_intersection(1, o0, d0, c, r, 0.001, 1.7976931348623157e308)
        if not (t < 1e300):
            continue
        p = o0 + t * d0                                   # hit.p = ray.origin + closest_t * ray.direction (main.odin:553)
        n = (p - c) / r
        reach = float(np.linalg.norm(np.float32(c).astype(np.float64)) + abs(np.float32(r))) * 1.000001
        prev = oracle.mixed_filter(o0, d0, c, r, reach)
        # outgoing direction: mostly away from the sphere, often almost tangent
        tang = rng.normal(size=3)
        tang -= n * (tang @ n)
        tang /= np.linalg.norm(tang)
        out = tang + n * rng.choice([1.0, 1e-2, 1e-5, 1e-8, 1e-11, -1e-9, 0.0]) * rng.uniform(0.1, 1.0)
        out *= 10.0 ** rng.uniform(-6, 2) / np.linalg.norm(out)
        f = oracle.mixed_filter(p, out, c, r, reach, start_cc_bound=prev["next_cc_bound"])
        t2 = oracle.ray_sphere_intersection(1, p, out, c, r, 0.001, 1.7976931348623157e308)
        hit = t2 < 1e300
        if f["kept"]:
            kept += 1
            hits_kept += hit
        else:
            dropped += 1
            assert not hit, (c, r, o0, d0, out, t2)
    print(f"start-sphere rule: dropped {dropped}, kept {kept} (of which the reference reports a hit for {hits_kept})")
    assert dropped > 500 and hits_kept > 0


def test_out_of_range_rays_keep_every_sphere(oracle):
    c, r = (0.0, 0.0, -1.0), 0.5
    for o, d in (((0.0, 0.0, 0.0), (0.0, 0.0, -1e-25)), ((0.0, 0.0, 0.0), (0.0, 3e21, 0.0)), ((2e7, 0.0, 0.0), (0.0, 1.0, 0.0)),
                 ((float("nan"), 0.0, 0.0), (0.0, 1.0, 0.0)), ((0.0, 0.0, 0.0), (float("inf"), 1.0, 0.0))):
        f = oracle.mixed_filter(o, d, c, r, 1.5)
        assert not f["in_range"] and f["kept"]


@pytest.mark.parametrize("name,width,spp", [("dev", 160, 12), ("two", 160, 12), ("final", 240, 4), ("stress", 128, 2)])
def test_mixed_render_is_the_f64_render_and_no_hit_sphere_is_dropped(rt, oracle, name, width, spp):
    if name == "stress":
        cs, sph, mat = rt.build_stress_scene(5, 2000, width)
    else:
        cs, sph, mat = {"dev": rt.build_dev_scene, "two": rt.build_two_sphere_scene, "final": lambda w: rt.build_final_scene(1, w)}[name](width)
    cs.samples_per_pixel = spp
    cam = rt.calculate_camera_render_data(cs)
    w, h = cs.image_width, cs.image_height
    a = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_F64, seed=5, threads=8)
    b = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_MIXED, seed=5, threads=8)
    assert int(b["stats"][6]) == 0                       # spheres dropped by the filter although the f64 test hits them
    assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["accum"], b["accum"])
    assert int(a["stats"][0]) == int(b["stats"][0])
    tests = int(a["stats"][0]) * len(sph)
    print(f"{name}: the filter keeps {int(b['stats'][5]) / tests:.4%} of {tests} ray/sphere tests")
    if len(sph) > 100:
        assert int(b["stats"][5]) * 50 < tests           # and it filters: under 2 % of the tests reach f64


def test_far_camera_and_scaled_worlds(rt, oracle):
    """a camera 5000 units away (|o| >> scene) and worlds scaled to the edges of the filter's range"""
    cs, sph, mat = rt.build_dev_scene(64)
    cs.samples_per_pixel = 4
    for scale in (1.0, 3e-5, 2e4, 1e-7, 1e7):   # the last two leave the range: the filter is not used
        s2 = sph.copy()
        s2["center"] *= scale
        s2["radius"] *= scale
        c2 = rt.CameraSettings.from_buffer_copy(cs)
        c2.position = rt.V3(*(np.array(cs.position.tuple()) * scale))
        c2.focus_distance = cs.focus_distance * scale
        cam = rt.calculate_camera_render_data(c2)
        w, h = c2.image_width, c2.image_height
        a = oracle.render_region(w, h, (0, 0, w, h), cam, s2, mat, precision=rt.PRECISION_F64, seed=9, threads=4)
        b = oracle.render_region(w, h, (0, 0, w, h), cam, s2, mat, precision=rt.PRECISION_MIXED, seed=9, threads=4)
        assert int(b["stats"][6]) == 0 and np.array_equal(a["image"], b["image"]), scale
    far = rt.CameraSettings.from_buffer_copy(cs)
    eye = np.array([-2000.0, 2000.0, 1000.0]) * 1.7
    far.position = rt.V3(*eye)
    right, up, fwd = rt.lookat(tuple(eye), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0))
    far.right, far.up, far.forward = rt.V3(*right), rt.V3(*up), rt.V3(*fwd)
    far.focus_distance = float(np.linalg.norm(eye - np.array([0.0, 0.0, -1.0])))
    far.vfov = 0.0002
    far.depth_of_field_angle = 0.0
    cam = rt.calculate_camera_render_data(far)
    w, h = far.image_width, far.image_height
    a = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_F64, seed=9, threads=4)
    b = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_MIXED, seed=9, threads=4)
    assert int(b["stats"][6]) == 0 and np.array_equal(a["image"], b["image"])
    assert int(a["stats"][0]) > w * h * 4 * 1.2      # the camera does see the spheres


def test_adversarial_worlds_audit(rt, oracle):
    """the CPU audit of the filter (every dropped sphere re-tested in f64) on worlds of extreme proportions: a ground 300
    world sizes large under tiny spheres, a negative radius, cameras near and three orders of magnitude away"""
    rng = np.random.default_rng(99)
    for case in range(12):
        scale = 10.0 ** rng.uniform(-3.0, 4.0)
        n = int(rng.choice([8, 40]))
        mat = np.zeros(4, dtype=rt.MATERIAL_DTYPE)
        mat["type"] = [0, 1, 2, 2]
        mat["albedo"] = rng.uniform(0.1, 0.95, size=(4, 3))
        mat["param1"] = [0.0, 0.3, 1.5, 1.0 / 1.5]
        sph = np.zeros(n, dtype=rt.SPHERE_DTYPE)
        sph["center"] = rng.normal(size=(n, 3)) * scale
        sph["radius"] = scale * 10.0 ** rng.uniform(-2.5, 0.3, size=n)
        sph["material_index"] = rng.integers(0, 4, size=n)
        sph["radius"][3] = -sph["radius"][3]
        sph["center"][4] = (0.0, -300.0 * scale, 0.0)
        sph["radius"][4] = 298.0 * scale
        cs, _, _ = rt.build_dev_scene(24)
        dist = scale * 10.0 ** rng.uniform(-0.5, 3.0)
        eye = rng.normal(size=3)
        eye *= dist / np.linalg.norm(eye)
        cs.position = rt.V3(*eye)
        right, up, fwd = rt.lookat(tuple(eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        cs.right, cs.up, cs.forward = rt.V3(*right), rt.V3(*up), rt.V3(*fwd)
        cs.focus_distance = float(dist)
        cs.vfov = float(min(0.3, 2.5 * scale / dist / 6.28))
        cs.depth_of_field_angle = 0.0
        cs.samples_per_pixel = 4
        cam = rt.calculate_camera_render_data(cs)
        w, h = cs.image_width, cs.image_height
        a = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_F64, seed=case, threads=4)
        b = oracle.render_region(w, h, (0, 0, w, h), cam, sph, mat, precision=rt.PRECISION_MIXED, seed=case, threads=4)
        assert int(b["stats"][6]) == 0 and np.array_equal(a["image"], b["image"]), (case, scale, dist)
