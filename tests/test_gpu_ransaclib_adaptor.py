"""The reference's OWN LO-MSAC driver (ransac_lib::LocallyOptimizedMSAC, lib/RansacLib/RansacLib/ransac.h:118-428) run over
the product's RansacLib Solver-concept adaptors (ppsfm/ransaclib_solvers.hpp) on the device.

oracle/_ref/ransaclib_adaptor is compiled in the build container from the reference's std-only headers where they lie
(oracle/Makefile, never copied) + the adaptors + libppsfm_hip.so; the binary travels to the GPU box.  Compared with the
library's own replay of that driver (pp_planar_lomsac / pp_pose2d_lomsac / pp_fourview2d_lomsac): with measurement noise
(scores separated beyond round-off) the two runs must take the same trajectory - iterations, LO runs, inlier set.
Reference instantiations: src/init/initializer.cc:119-123, 201-206; src/init/sfm2d_test.cc:177-183."""
import os
import subprocess

import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ransaclib_adaptor")


def _run(mode, numbers, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/ransaclib_adaptor was not built (needs /root/reference at build time)")
    path = str(tmp_path / (mode + ".txt"))
    with open(path, "w") as f:
        f.write("\n".join(repr(float(v)) if isinstance(v, (float, np.floating)) else str(int(v)) for v in numbers))
    out = subprocess.run([EXE, mode, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        key, _, rest = line.partition(" ")
        rows[key] = rest.split()
    ninl, iters, best_inl, lo, score, ratio = rows["stats"]
    return dict(ninl=int(ninl), iterations=int(iters), best_inliers=int(best_inl), lo=int(lo), score=float(score), ratio=float(ratio),
                inliers=np.array(rows["inliers"], dtype=np.int64), cams=np.array(rows["cams"], dtype=np.float64))


def _opts(min_it, max_it, thr, seed=0, final_lsq=0):
    return [min_it, max_it, float(thr), seed, final_lsq]


@pytest.mark.parametrize("n,nout,seed,noise", [(100, 20, 8, 1e-4), (400, 120, 9, 2e-4)])
def test_reference_driver_over_planar_offset_solver(tmp_path, n, nout, seed, noise):
    from privacy_preserving_sfm_amd.device import PlanarOffsetProblem, lomsac_options
    sc = synthetic.make_planar_offset_scene(n, n_outliers=nout, seed=seed, noise=noise)
    thr = 0.005
    nums = [n] + [float(v) for v in np.asarray(sc["poses"]).ravel()] + [float(v) for v in np.asarray(sc["Rg"]).ravel()]
    nums += [float(v) for v in np.asarray(sc["lines"]).ravel()] + _opts(100, 10000, thr)
    ref = _run("planar", nums, tmp_path)
    pp = PlanarOffsetProblem(sc["poses"], sc["lines"], sc["Rg"])
    rep, off, cams, idx = pp.lomsac(lomsac_options(squared_inlier_threshold=thr))
    pp.close()
    assert ref["ninl"] == ref["best_inliers"] == rep.best_num_inliers >= n - nout - 2
    assert ref["iterations"] == rep.num_iterations and ref["lo"] == rep.number_lo_iterations
    assert np.array_equal(ref["inliers"], idx)
    assert np.allclose(ref["cams"].reshape(4, 3, 4), cams, rtol=1e-9, atol=1e-12)
    assert abs(ref["score"] - rep.best_model_score) <= 1e-12 * rep.best_model_score       # sequential vs fixed-tree MSAC sum
    assert abs(ref["ratio"] - rep.inlier_ratio) < 1e-15


@pytest.mark.parametrize("n,nout,seed,thr,noise", [(100, 20, 4, 2e-5, 1e-3), (400, 150, 9, 2e-6, 5e-4)])
def test_reference_driver_over_absolute_pose2d_solver(tmp_path, n, nout, seed, thr, noise):     # sfm2d_test.cc:164-236
    from privacy_preserving_sfm_amd.device import Pose2dProblem, lomsac_options
    sc = synthetic.make_scene_2d(4, n, n_outliers=nout, seed=seed)
    rng = np.random.default_rng(seed)
    x = sc["x"][1] + noise * rng.normal(size=sc["x"][1].shape)
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    nums = [n] + [float(v) for v in x.ravel()] + [float(v) for v in np.asarray(sc["X"]).ravel()] + _opts(100, 10000, thr)
    ref = _run("pose2d", nums, tmp_path)
    pp = Pose2dProblem(x, sc["X"])
    rep, pose, idx = pp.lomsac(lomsac_options(squared_inlier_threshold=thr))
    pp.close()
    assert ref["best_inliers"] == rep.best_num_inliers and ref["iterations"] == rep.num_iterations and ref["lo"] == rep.number_lo_iterations
    assert np.array_equal(ref["inliers"], idx)
    assert np.allclose(ref["cams"].reshape(2, 3), pose, rtol=1e-9, atol=1e-12)


def test_reference_driver_over_fourview2d_solver(tmp_path):
    """FourView2dEstimator under the reference's driver (initializer.cc:119-123 with final_least_squares): MinimalSolver (16
    candidates per sample), NonMinimalSolver, LeastSquares (the two restated LM problems) all go through the adaptor.
    The LO-MSAC replay in the library batches the same calls; the run is compared on its outcome and its trajectory."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem, fourview2d_default_frames, lomsac_options
    n, nout, noise, thr = 120, 30, 2e-4, 2e-3
    sc = synthetic.make_scene_2d(4, n, n_outliers=nout, seed=8)
    rng = np.random.default_rng(8)
    x = sc["x"] + noise * rng.normal(size=sc["x"].shape)
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    frames = fourview2d_default_frames()
    nums = [n] + [float(v) for v in x.ravel()] + [float(v) for v in frames] + _opts(100, 10000, thr, final_lsq=1)
    ref = _run("fourview2d", nums, tmp_path)
    fv = FourView2dProblem(x)
    rep, cams, X, idx = fv.lomsac(lomsac_options(squared_inlier_threshold=thr, final_least_squares=1))
    fv.close()
    assert ref["best_inliers"] >= n - nout - n // 10 and sc["is_outlier"][ref["inliers"]].sum() <= 2
    assert ref["iterations"] == rep.num_iterations and ref["lo"] == rep.number_lo_iterations
    assert ref["best_inliers"] == rep.best_num_inliers and np.array_equal(ref["inliers"], idx)
    assert np.abs(ref["cams"].reshape(4, 2, 3) - cams).max() < 1e-7
