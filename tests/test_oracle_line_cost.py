"""Oracle (jets) vs the 50-digit sympy/mpmath golden vectors for the line residual + Jacobian.

Pins oracle/line_cost.h + oracle/camera_models.h (reference src/base/cost_functions.h:62-100,
src/base/camera_models.h) on all 11 camera models, including the three FOV branches.
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "line_cost_golden.json")


def _cases():
    with open(GOLD) as f:
        return json.load(f)["cases"]


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def test_golden_covers_all_models():
    models = {c["model"] for c in _cases()}
    assert models == set(range(11))
    assert {c["branch"] for c in _cases() if c["model"] == 7} == {0, 1, 2}


@pytest.mark.parametrize("idx", range(69))
def test_oracle_matches_golden(oracle, idx):
    c = _cases()[idx]
    r, Jq, Jt, JX, Jc = oracle.line_cost(c["model"], c["line"], c["q"], c["t"], c["X"], c["cam"])
    # residual is a difference of two pixel coordinates (~1e3): absolute floor 1e-9 px
    assert _close(r, c["r"], 1e-10, 2e-9), (c["name"], r, c["r"])
    scale = max(1.0, np.abs(np.array(c["Jq"])).max(), np.abs(np.array(c["JX"])).max())
    for name, got, want in (("Jq", Jq, c["Jq"]), ("Jt", Jt, c["Jt"]), ("JX", JX, c["JX"]), ("Jcam", Jc, c["Jcam"])):
        assert _close(got, want, 1e-9, 1e-9 * scale), (c["name"], name, got, want)


def test_rank_one_residual_for_undistorted_models(oracle):
    # for (SIMPLE_)PINHOLE r = (fx*alpha*a, fy*alpha*b)
    rng = np.random.default_rng(1)
    for model, cam in ((0, [900.0, 640, 480]), (1, [900.0, 950.0, 640, 480])):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = np.array([0.1, -0.2, 5.0]); X = rng.uniform(-1, 1, 3)
        th = 0.7; line = np.array([np.cos(th), np.sin(th), 0.05])
        r, *_ = oracle.line_cost(model, line, q, t, X, cam)
        # recompute alpha independently
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        p = R @ X + t
        alpha = line[0] * p[0] / p[2] + line[1] * p[1] / p[2] + line[2]
        fx, fy = (cam[0], cam[0]) if model == 0 else (cam[0], cam[1])
        assert np.allclose(r, [fx * alpha * line[0], fy * alpha * line[1]], rtol=1e-9, atol=1e-9)
