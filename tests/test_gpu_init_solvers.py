"""K6 parity: planar-offset minimal solver / triangulate-all + score / LO-MSAC, and the four-view 2D
triangulate-all + score, against the oracle.

Tolerances: the device triangulates by precomputed pseudo-inverse / 2x2 normal equations, the oracle by Householder
QR (what Eigen's colPivHouseholderQr().solve computes): both are the least-squares solution, agreement 1e-9
relative.  The MSAC score is summed in index order on both sides; the LO-MSAC trajectory (iterations, LO count,
inlier set) is identical on scenes whose errors are separated from the threshold.
"""
import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,nout", [(20, 0), (257, 40)])
def test_planar_solver_and_score_match_oracle(oracle, n, nout):
    from privacy_preserving_sfm_amd.device import PlanarOffsetProblem
    sc = synthetic.make_planar_offset_scene(n, n_outliers=nout, seed=10 + n, noise=1e-4 if nout else 0.0)
    pp = PlanarOffsetProblem(sc["poses"], sc["lines"], sc["Rg"])
    rng = np.random.default_rng(0)
    samples = np.stack([rng.choice(n, 3, replace=False) for _ in range(200)]).astype(np.int32)
    off = pp.solve_batch(samples)
    ref = oracle.planar_minimal(sc, samples)
    good = np.isfinite(ref).all(axis=1) & (np.abs(ref).max(axis=1) < 1e3)
    assert good.sum() > 150
    assert np.allclose(off[good], ref[good], rtol=1e-7, atol=1e-9)
    # non-minimal (least squares) variant used by the local optimisation, 20 samples
    s20 = np.stack([rng.choice(n, 20, replace=False) for _ in range(8)]).astype(np.int32)
    assert np.allclose(pp.solve_batch(s20), oracle.planar_minimal(sc, s20), rtol=1e-7, atol=1e-9)
    thr = 0.005
    models = np.concatenate([sc["t_gt"][None], off[good][:40], sc["t_gt"][None] + 0.3])
    score, inl = pp.score(models, thr)
    for m in range(len(models)):
        rs, ri, rerr, rX = oracle.planar_score(sc, models[m], thr)
        err, X, cams = pp.evaluate(models[m])
        far = np.abs(rerr - thr) > 1e-9            # tracks whose error is not within rounding of the threshold
        assert np.allclose(err[far], rerr[far], rtol=1e-7, atol=1e-10)
        assert inl[m] == ri
        assert abs(score[m] - rs) <= 1e-9 * rs + 1e-11
        assert np.array_equal(err == 100000.0, rerr == 100000.0)
    err, X, cams = pp.evaluate(sc["t_gt"])
    assert np.abs(cams - sc["gt_cams"]).max() < 1e-12 and np.median(np.abs(X[~sc["is_outlier"]] - sc["X"][~sc["is_outlier"]])) < 1e-2
    pp.close()


@pytest.mark.parametrize("n,nout,seed,noise", [(20, 0, 7, 0.0), (20, 0, 7, 1e-4), (100, 20, 8, 1e-4), (1000, 300, 9, 2e-4)])
def test_planar_lomsac_matches_oracle(oracle, n, nout, seed, noise):
    """initializer_test.cc:234-341 shapes + a larger one.  With noise-free data every all-inlier sample scores
    ~1e-15 (pure rounding), so `score < best` is decided by rounding noise and only the RESULT is compared; with
    measurement noise the scores are separated and the whole LO-MSAC trajectory must coincide."""
    from privacy_preserving_sfm_amd.device import PlanarOffsetProblem, lomsac_options
    sc = synthetic.make_planar_offset_scene(n, n_outliers=nout, seed=seed, noise=noise)
    pp = PlanarOffsetProblem(sc["poses"], sc["lines"], sc["Rg"])
    thr = 0.005 * 0.005 if noise == 0.0 else 0.005
    rep, off, cams, idx = pp.lomsac(lomsac_options(squared_inlier_threshold=thr))
    inl, rcams, st, ridx = oracle.planar_lomsac(sc, oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=thr))
    assert rep.best_num_inliers == inl >= n - nout
    assert np.array_equal(idx, ridx)
    assert rep.hypotheses_evaluated >= rep.num_iterations
    if noise == 0.0:
        assert np.abs(cams - sc["gt_cams"]).max() < 1e-7 and np.abs(rcams - sc["gt_cams"]).max() < 1e-7
    else:
        assert rep.num_iterations == st.num_iterations and rep.number_lo_iterations == st.number_lo_iterations
        assert np.allclose(cams, rcams, rtol=1e-7, atol=1e-9)
        assert abs(rep.best_model_score - st.best_model_score) <= 1e-9 * st.best_model_score
        assert np.abs(cams - sc["gt_cams"]).max() < 5e-2
    pp.close()


def test_fourview2d_score_matches_oracle(oracle):
    from privacy_preserving_sfm_amd.device import FourView2dProblem
    sc = synthetic.make_scene_2d(4, 333, n_outliers=60, seed=12)
    fv = FourView2dProblem(sc["x"] * 3.0)            # the constructor normalises the bearings (sfm2d.h:62-67)
    rng = np.random.default_rng(1)
    models = [sc["cams"]]
    for k in range(20):
        c = sc["cams"].copy(); c[1:] += rng.normal(0, 10.0 ** rng.uniform(-6, -1), (3, 2, 3)); models.append(c)
    c = sc["cams"].copy(); c[2] = -c[2]; models.append(c)
    models = np.array(models)
    thr = 1e-4
    score, inl = fv.score(models, thr)
    for m in range(len(models)):
        rs, ri, rerr, rX = oracle.fourview2d_score(models[m], sc["x"], thr)
        err, X = fv.evaluate(models[m])
        far = np.abs(rerr - thr) > 1e-10
        assert np.allclose(err[far], rerr[far], rtol=1e-7, atol=1e-11)
        assert inl[m] == ri and abs(score[m] - rs) <= 1e-9 * rs + 1e-11
    assert inl[0] == 333 - 60 and inl[-1] == 0
    fv.close()
