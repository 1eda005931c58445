"""The C++ host mirror (ppsfm/ppsfm.hpp) EXECUTED on the device (it had only ever been compiled).

tests/cpp_mirror_gpu_test.cpp is built with g++ against libppsfm_hip.so and run as a child process on scenes written
to a text file; its output is compared with the oracle (P6L RANSAC, BA) and with the reference tests' bounds
(four-view initialisation, src/init/initializer_test.cc:346-435).  Reference callers: src/estimators/pose.cc:48-94,
src/optim/bundle_adjustment.cc:260-320, src/init/initializer.cc:58-216."""
import os
import subprocess

import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from privacy_preserving_sfm_amd import build
    out = str(tmp_path_factory.mktemp("cpp") / "cpp_mirror_gpu_test")
    libdir = os.path.dirname(build.LIB)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-o", out, os.path.join(ROOT, "tests", "cpp_mirror_gpu_test.cpp"),
                           "-L" + libdir, "-lppsfm_hip", "-Wl,-rpath," + libdir])
    return out


def _run(exe, mode, numbers, tmp_path):
    path = str(tmp_path / (mode + ".txt"))
    with open(path, "w") as f:
        f.write("\n".join(repr(float(v)) if isinstance(v, (float, np.floating)) else str(int(v)) for v in numbers))
    out = subprocess.run([exe, mode, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        key, _, rest = line.partition(" ")
        rows.setdefault(key, []).append(rest.split())
    return rows


def _flat(*arrays):
    out = []
    for a in arrays:
        a = np.asarray(a)
        out += [x for x in (a.astype(np.float64).ravel() if a.dtype.kind == "f" else a.astype(np.int64).ravel())]
    return out


def test_cpp_estimate_absolute_pose_from_lines(exe, tmp_path, oracle):
    sc = synthetic.make_ransac_scene(600, outlier_ratio=0.35, noise_px=0.3, seed=77, aligned_ratio=0.2)
    kw = dict(min_inlier_ratio=0.25, confidence=0.99999, min_num_trials=100, max_num_trials=10000)
    nums = [600, float(sc["max_error"]), 0.25, 0.99999, 3.0, 100, 10000, 0]
    for i in range(600):
        nums += [float(v) for v in sc["lines"][i]] + [int(sc["aligned"][i])] + [float(v) for v in sc["points"][i]]
    rows = _run(exe, "pose", nums, tmp_path)
    ref, ref_mask = oracle.p6l_ransac(sc["lines"], sc["points"], sc["aligned"], sc["max_error"], seed=0, mult=3.0, **kw)
    success, trials, ninl, rsum = rows["report"][0]
    assert int(success) == 1 and int(trials) == ref.num_trials and int(ninl) == ref.num_inliers
    assert abs(float(rsum) - ref.residual_sum) <= 1e-6 * ref.residual_sum
    mask = np.array([int(c) for c in rows["mask"][0][0]], dtype=np.uint8)
    assert np.array_equal(mask, ref_mask)                                               # inlier set bit-identical
    model = np.array(rows["model"][0], dtype=np.float64)
    assert np.allclose(model, np.array(ref.model), rtol=1e-7, atol=1e-8)
    ok, ninl2 = rows["pose"][0]
    assert int(ok) == 1 and int(ninl2) == ref.num_inliers
    q = np.array(rows["qvec"][0], dtype=np.float64); t = np.array(rows["tvec"][0], dtype=np.float64)
    R = synthetic.quat_to_rot(q)
    assert np.abs(R - model.reshape(3, 4)[:, :3]).max() < 1e-9 and np.array_equal(t, model.reshape(3, 4)[:, 3])
    assert np.abs(R - sc["gt_pose"][:, :3]).max() < 2e-2
    # Estimator concept
    want = oracle.p6l(sc["lines"][:6], sc["points"][:6], sc["aligned"][:6])
    assert int(rows["p6l"][0][0]) == len(want)
    for got, w in zip(rows.get("p6l_model", []), want):
        assert np.allclose(np.array(got, dtype=np.float64).reshape(3, 4), w, rtol=1e-5, atol=1e-5)
    res = np.array(rows["residuals"][0], dtype=np.float64)
    assert np.array_equal(res, oracle.line_residuals(sc["lines"], sc["points"], model.reshape(3, 4)))     # bit-exact residuals


def test_cpp_bundle_adjustment_problem_solve(exe, tmp_path, oracle):
    sc = synthetic.make_ba_scene(20, 500, 4, seed=0xC0FFEE + 1, model=2)
    C, P, K, M = 20, 500, len(sc["camera_model"]), len(sc["obs_pose"])
    nums = [C, P, K, M, 0, 1.0]
    nums += _flat(sc["lines"], sc["obs_pose"], sc["obs_point"], sc["pose_camera"], sc["camera_model"], sc["pose_const"], sc["tvec_const_mask"],
                  sc["point_const"], np.full(K, 0xFFFF, dtype=np.int64), sc["poses"], sc["points"], np.asarray(sc["intr"], dtype=np.float64))
    nums += [5, 0.0]        # five iterations: all above the rounding level of this noise-free scene
    rows = _run(exe, "ba", nums, tmp_path)
    rposes, rpoints, _, rs, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=5))
    usable, term, iters, succ, nres, c0, c1, calls = rows["summary"][0]
    assert int(usable) == 1 and int(iters) == rs.num_iterations and int(succ) == rs.num_successful_steps and int(nres) == 2 * M
    assert int(calls) == rs.num_iterations + 1                                           # the iteration callback ran (iteration 0 included)
    assert abs(float(c0) - rs.initial_cost) <= 1e-10 * rs.initial_cost
    poses = np.array(rows["poses"][0], dtype=np.float64).reshape(C, 7)
    points = np.array(rows["points"][0], dtype=np.float64).reshape(P, 3)
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max()
    assert np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()


@pytest.mark.parametrize("nout,tol,seed", [(0, 1e-6, 3), (10, 1e-4, 4)])
def test_cpp_initialize_reconstruction(exe, tmp_path, nout, tol, seed):
    from privacy_preserving_sfm_amd.initializer import InitOptions, initialize_reconstruction
    sc = synthetic.make_init_scene(100, 50, n_outliers=nout, seed=seed)
    nums = []
    for v in range(4):
        nums += [float(x) for x in sc["gravity"][v]] + [len(sc["lines"][v])]
        for l, a in zip(sc["lines"][v], sc["aligned"][v]):
            nums += [float(x) for x in l] + [int(a)]
    rows = _run(exe, "init", nums, tmp_path)
    ok, ratio, n = rows["init"][0]
    assert int(ok) == 1 and int(n) == 4
    poses = np.array(rows["pose"], dtype=np.float64).reshape(4, 3, 4)
    # identical to the Python mirror (same C-ABI calls, same seeds) ...
    ok_py, poses_py, ratio_py = initialize_reconstruction(sc["lines"], sc["aligned"], sc["gravity"], InitOptions())
    assert ok_py and np.allclose(poses, poses_py, rtol=1e-9, atol=1e-12) and abs(float(ratio) - ratio_py) < 1e-12
    # ... and within the reference test's bound of the ground truth
    poses[:, :, 3] /= np.linalg.norm(poses[1][:, 3])
    for i in range(4):
        assert np.linalg.norm(poses[i] - sc["cams"][i]) < tol
