#!/usr/bin/env python3
"""Generate tests/golden/line_cost_golden.json.

Independent high-precision check of the line-to-point residual and its Jacobian
(reference src/base/cost_functions.h:62-100 with the 11 camera models of
src/base/camera_models.h).  The reference ships NO golden vectors for this path
(SURVEY.md §4), and Ceres/Eigen are not installed, so the vectors are minted here:
the residual is written symbolically with sympy (straight from the formulas, not from
the oracle's C++), differentiated symbolically, and evaluated with 50-digit mpmath
arithmetic; results are rounded to double.  The oracle (jets) and the HIP kernel
(analytic chain rule) are both tested against these numbers.

Run from the repo root:  python tests/golden/gen_line_cost_golden.py
"""
import json
import os
import random

import mpmath as mp
import sympy as sp

mp.mp.dps = 50

NUM_PARAMS = [3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12]
NAMES = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE",
         "FULL_OPENCV", "FOV", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"]


def fisheye(u, v, ks):
    r = sp.sqrt(u * u + v * v)
    th = sp.atan(r)
    series = 1
    for i, k in enumerate(ks):
        series = series + k * th ** (2 * (i + 1))
    thd = th * series
    return u * thd / r - u, v * thd / r - v


def world_to_image(model, p, u, v, branch):
    """branch selects the piece of a piecewise model (FOV: 0 small omega, 1 small radius, 2 generic)."""
    if model == 0:
        return p[0] * u + p[1], p[0] * v + p[2]
    if model == 1:
        return p[0] * u + p[2], p[1] * v + p[3]
    if model == 2:
        r2 = u * u + v * v
        rad = p[3] * r2
        return p[0] * (u + u * rad) + p[1], p[0] * (v + v * rad) + p[2]
    if model == 3:
        r2 = u * u + v * v
        rad = p[3] * r2 + p[4] * r2 * r2
        return p[0] * (u + u * rad) + p[1], p[0] * (v + v * rad) + p[2]
    if model == 4:
        u2, uv, v2 = u * u, u * v, v * v
        r2 = u2 + v2
        rad = p[4] * r2 + p[5] * r2 * r2
        du = u * rad + 2 * p[6] * uv + p[7] * (r2 + 2 * u2)
        dv = v * rad + 2 * p[7] * uv + p[6] * (r2 + 2 * v2)
        return p[0] * (u + du) + p[2], p[1] * (v + dv) + p[3]
    if model == 5:
        du, dv = fisheye(u, v, [p[4], p[5], p[6], p[7]])
        return p[0] * (u + du) + p[2], p[1] * (v + dv) + p[3]
    if model == 6:
        u2, uv, v2 = u * u, u * v, v * v
        r2 = u2 + v2
        r4 = r2 * r2
        r6 = r4 * r2
        rad = (1 + p[4] * r2 + p[5] * r4 + p[8] * r6) / (1 + p[9] * r2 + p[10] * r4 + p[11] * r6)
        du = u * rad + 2 * p[6] * uv + p[7] * (r2 + 2 * u2) - u
        dv = v * rad + 2 * p[7] * uv + p[6] * (r2 + 2 * v2) - v
        return p[0] * (u + du) + p[2], p[1] * (v + dv) + p[3]
    if model == 7:
        om = p[4]
        rad2 = u * u + v * v
        om2 = om * om
        if branch == 0:
            f = (om2 * rad2) / 3 - om2 / 12 + 1
        elif branch == 1:
            th = sp.tan(om / 2)
            f = (-2 * th * (4 * rad2 * th * th - 3)) / (3 * om)
        else:
            rad = sp.sqrt(rad2)
            f = sp.atan(rad * 2 * sp.tan(om / 2)) / (rad * om)
        return p[0] * (u * f) + p[2], p[1] * (v * f) + p[3]
    if model == 8:
        du, dv = fisheye(u, v, [p[3]])
        return p[0] * (u + du) + p[1], p[0] * (v + dv) + p[2]
    if model == 9:
        du, dv = fisheye(u, v, [p[3], p[4]])
        return p[0] * (u + du) + p[1], p[0] * (v + dv) + p[2]
    if model == 10:
        r = sp.sqrt(u * u + v * v)
        th = sp.atan(r)
        uu, vv = th * u / r, th * v / r
        u2, uv, v2 = uu * uu, uu * vv, vv * vv
        r2 = u2 + v2
        r4 = r2 * r2
        r6 = r4 * r2
        r8 = r6 * r2
        rad = p[4] * r2 + p[5] * r4 + p[8] * r6 + p[9] * r8
        du = uu * rad + 2 * p[6] * uv + p[7] * (r2 + 2 * u2) + p[10] * r2
        dv = vv * rad + 2 * p[7] * uv + p[6] * (r2 + 2 * v2) + p[11] * r2
        return p[0] * (uu + du) + p[2], p[1] * (vv + dv) + p[3]
    raise ValueError(model)


def build(model, branch):
    n = NUM_PARAMS[model]
    q = sp.symbols("q0:4")
    t = sp.symbols("t0:3")
    X = sp.symbols("X0:3")
    cam = sp.symbols("c0:%d" % n)
    a, b, c = sp.symbols("la lb lc")
    qv = sp.Matrix(q[1:])
    Xv = sp.Matrix(X)
    uv_ = 2 * qv.cross(Xv)
    p = Xv + q[0] * uv_ + qv.cross(uv_) + sp.Matrix(t)
    u, v = p[0] / p[2], p[1] / p[2]
    alpha = a * u + b * v + c
    fu, fv = u - alpha * a, v - alpha * b
    x, y = world_to_image(model, cam, u, v, branch)
    xf, yf = world_to_image(model, cam, fu, fv, branch)
    r = sp.Matrix([x - xf, y - yf])
    params = list(q) + list(t) + list(X) + list(cam)
    J = r.jacobian(params)
    syms = params + [a, b, c]
    fr = sp.lambdify(syms, r, "mpmath")
    fJ = sp.lambdify(syms, J, "mpmath")
    return fr, fJ, n


def sample(rng, model, kind):
    n = NUM_PARAMS[model]
    # unit quaternion, point in front of the camera
    ax = [rng.uniform(-1, 1) for _ in range(3)]
    ang = rng.uniform(-0.6, 0.6)
    nrm = sum(x * x for x in ax) ** 0.5
    import math
    q = [math.cos(ang / 2)] + [math.sin(ang / 2) * x / nrm for x in ax]
    qn = sum(x * x for x in q) ** 0.5
    q = [x / qn for x in q]
    t = [rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(3.0, 6.0)]
    X = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)]
    f = rng.uniform(800, 1200)
    two_f = n in (4, 8, 12, 5) and model not in (2, 3, 8, 9)
    if model in (0, 2, 3, 8, 9):
        cam = [f, 640.0 + rng.uniform(-5, 5), 480.0 + rng.uniform(-5, 5)]
    else:
        cam = [f, f * rng.uniform(0.95, 1.05), 640.0 + rng.uniform(-5, 5), 480.0 + rng.uniform(-5, 5)]
    extra = n - len(cam)
    if model == 7:
        if kind == 0:
            cam.append(rng.uniform(0.001, 0.009))      # omega^2 < 1e-4
        else:
            cam.append(rng.uniform(0.2, 0.9))
    else:
        cam += [rng.uniform(-0.05, 0.05) for _ in range(extra)]
    th = rng.uniform(0, 2 * math.pi)
    a, b = math.cos(th), math.sin(th)
    c = rng.uniform(-0.3, 0.3)
    if model == 7 and kind == 1:
        # force tiny radius: put the point on the optical axis, line through the centre
        X = [0.0, 0.0, 0.5]
        q = [1.0, 0.0, 0.0, 0.0]
        t = [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 4.0]
        c = rng.uniform(-2e-3, 2e-3)
    del two_f
    return q, t, X, cam, [a, b, c]


def main():
    rng = random.Random(20260928)
    cases = []
    for model in range(11):
        variants = [(2, 2)] if model != 7 else [(0, 0), (1, 1), (2, 2)]
        for branch, kind in variants:
            fr, fJ, n = build(model, branch)
            for _ in range(6 if model != 7 else 3):
                q, t, X, cam, line = sample(rng, model, kind)
                args = [mp.mpf(x) for x in (q + t + X + cam + line)]
                r = fr(*args)
                J = fJ(*args)
                r = [float(r[i]) for i in range(2)]
                Jm = [[float(J[i, j]) for j in range(10 + n)] for i in range(2)]
                cases.append({
                    "model": model, "name": NAMES[model], "branch": branch,
                    "q": q, "t": t, "X": X, "cam": cam, "line": line,
                    "r": r,
                    "Jq": [Jm[i][0:4] for i in range(2)],
                    "Jt": [Jm[i][4:7] for i in range(2)],
                    "JX": [Jm[i][7:10] for i in range(2)],
                    "Jcam": [Jm[i][10:10 + n] for i in range(2)],
                })
        print("model", model, "done", flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "line_cost_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/gen_line_cost_golden.py", "digits": 50, "cases": cases}, f)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
