"""The hierarchy's box test in the mixed mode may visit too much, never too little.

The traversal evaluates the slab test in f32 on the f64 ray rounded to f32 and widens it by a slack
(rt_kernel.hip, `meets_box` and `box_slack`).  A box that the segment {o + t d, 0 <= t <= t_max} of the f64 ray really
meets must pass that test, or a sphere inside it could be missed.  Checked here against exact rational arithmetic on
adversarial cases: rays aimed at corners, edges and faces of the box, origins up to 10^4 box sizes away, scenes far
from the coordinate origin, directions with tiny or zero components, t_max a hair before or behind the entry point.
The f32 side restates the kernel's operations with numpy float32 scalars and exactly rounded fused multiply-adds.
"""
from fractions import Fraction

import numpy as np

from pyref32 import F, fma

F32_MAX = np.finfo(np.float32).max
TWO60 = F(2.0 ** 60)


def fmax(a, b):  # v_max_f32: a NaN operand is ignored
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a >= b else b


def fmin(a, b):
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a <= b else b


def kernel_meets_box(o, d, t_max, origin, step, q_lo, q_hi):
    """the mixed instantiation's slab test: True = the box is entered"""
    with np.errstate(all="ignore"):
        fo = [F(x) for x in o]
        fd = [F(x) for x in d]
        A, B, OI = [], [], []
        for k in range(3):
            i = F(1.0) / fd[k]
            a, b, oi = F(step[k]) * i, (F(origin[k]) - fo[k]) * i, F(0.0)
            if not (abs(i) <= TWO60):
                a, b = F(0.0), F(np.nan)
            else:
                oi = abs(fo[k] * i)
            A.append(a); B.append(b); OI.append(oi)

        def slack_of(a, b, c):
            e = fma(F(65535.0), abs(a), abs(b)) + c
            return e if e < F32_MAX else F(0.0)

        slack = F(2.0 ** -20) * fmax(fmax(slack_of(A[0], B[0], OI[0]), slack_of(A[1], B[1], OI[1])), slack_of(A[2], B[2], OI[2]))
        t0 = [fma(F(q_lo[k]), A[k], B[k]) for k in range(3)]
        t1 = [fma(F(q_hi[k]), A[k], B[k]) for k in range(3)]
        t_up = F(t_max)
        if float(t_up) < t_max:
            t_up = np.nextafter(t_up, F(np.inf))  # __double2float_ru
        t_in = fmax(fmax(fmin(t0[0], t1[0]), fmin(t0[1], t1[1])), fmax(fmin(t0[2], t1[2]), F(0.0)))
        t_out = fmin(fmin(fmax(t0[0], t1[0]), fmax(t0[1], t1[1])), fmin(fmax(t0[2], t1[2]), t_up))
        return bool(t_in <= t_out + slack)


def exact_meets_box(o, d, t_max, origin, step, q_lo, q_hi):
    lo_t, hi_t = Fraction(0), Fraction(float(t_max))
    for k in range(3):
        lo = Fraction(float(F(origin[k]))) + q_lo[k] * Fraction(float(F(step[k])))
        hi = Fraction(float(F(origin[k]))) + q_hi[k] * Fraction(float(F(step[k])))
        ok, dk = Fraction(float(o[k])), Fraction(float(d[k]))
        if dk == 0:
            if not (lo <= ok <= hi):
                return False
            continue
        a, b = (lo - ok) / dk, (hi - ok) / dk
        lo_t, hi_t = max(lo_t, min(a, b)), min(hi_t, max(a, b))
    return lo_t <= hi_t


def test_box_test_never_skips_a_box_the_exact_segment_meets():
    rng = np.random.default_rng(20250311)
    met = skipped_ok = 0
    for case in range(2500):
        scale = 10.0 ** rng.uniform(-2, 3)
        origin = rng.normal(size=3) * scale
        if case % 2:      # the whole scene far from the coordinate origin: |o| is large next to |grid origin - o|
            origin = origin + rng.normal(size=3) * scale * 10.0 ** rng.uniform(1, 4)
        step = np.float32(scale * 10.0 ** rng.uniform(-5, -3.5) * rng.uniform(0.5, 2.0, size=3))
        q_lo = rng.integers(0, 65000, size=3)
        q_hi = q_lo + rng.integers(1, 500, size=3)
        q_hi = np.minimum(q_hi, 65535)
        lo = np.float64(np.float32(origin)) + q_lo * np.float64(step)
        hi = np.float64(np.float32(origin)) + q_hi * np.float64(step)
        # a point on the box: corner, edge, face or inside, then nudged by a relative 1e-9 .. 1e-16 either way
        w = rng.choice([0.0, 1.0, rng.uniform()], size=3)
        target = lo + w * (hi - lo)
        target = target + (hi - lo) * rng.choice([-1.0, 0.0, 1.0], size=3) * 10.0 ** rng.uniform(-16, -9)
        far = 10.0 ** rng.uniform(-1, 4) * float(np.max(hi - lo)) * 30.0
        o = target + rng.normal(size=3) * far
        kind = rng.integers(0, 4)
        if kind == 0:     # origin beside the box along one or two axes: the ray runs along a face
            axes = rng.choice(3, size=rng.integers(1, 3), replace=False)
            o[axes] = target[axes]
        d = target - o
        if kind == 1:     # a component that is tiny next to the others, or zero
            k = rng.integers(0, 3)
            d[k] = rng.choice([0.0, 1e-30, -1e-30, 1e-12]) * np.linalg.norm(d)
        length = np.linalg.norm(d)
        if length == 0.0:
            continue
        d = d * 10.0 ** rng.uniform(-3, 3) / length
        if not (2.0 ** -40 <= d @ d <= 2.0 ** 40 and o @ o <= 2.0 ** 40):
            continue      # such a ray visits every box (tested on the GPU)
        t_hit = float(np.linalg.norm(target - o) / np.linalg.norm(d))
        t_max = t_hit * (1.0 + rng.choice([-1.0, 1.0, 1e3]) * 10.0 ** rng.uniform(-15, -6)) if rng.uniform() < 0.8 else np.finfo(np.float64).max
        exact = exact_meets_box(o, d, t_max, origin, step, q_lo.tolist(), q_hi.tolist())
        if exact:
            met += 1
            assert kernel_meets_box(o, d, t_max, origin, step, q_lo, q_hi), (case, o, d, t_max, origin, step, q_lo, q_hi)
        elif not kernel_meets_box(o, d, t_max, origin, step, q_lo, q_hi):
            skipped_ok += 1
    assert met > 700 and skipped_ok > 100, (met, skipped_ok)   # both outcomes are exercised
