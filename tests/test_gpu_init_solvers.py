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


def _match_models(gpu16, ref16):
    """worst distance of any device candidate to its nearest oracle candidate (the 8 sign choices may come out in
    another order when a null vector's sign differs)"""
    worst = 0.0
    for g in gpu16:
        worst = max(worst, min(np.abs(g - r).max() / max(1.0, np.abs(r).max()) for r in ref16))
    return worst


@pytest.mark.parametrize("m", [5, 10])
def test_fourview2d_minimal_solver_matches_oracle(oracle, m):     # sfm2d.cc:363-444
    from privacy_preserving_sfm_amd.device import FourView2dProblem, fourview2d_default_frames
    sc = synthetic.make_scene_2d(4, 120, seed=41)
    x = sc["x"] + (0 if m == 5 else np.random.default_rng(2).normal(0, 1e-4, sc["x"].shape))
    x = x / np.linalg.norm(x, axis=2, keepdims=True)
    fv = FourView2dProblem(x)
    rng = np.random.default_rng(m)
    samples = np.stack([rng.choice(120, m, replace=False) for _ in range(300)]).astype(np.int32)
    frames = fourview2d_default_frames()
    for fr in (frames, rng.uniform(-1, 1, 12)):
        cams, cnt = fv.minimal_batch(samples, frames=fr)
        rcams, rcnt = oracle.fourview2d_minimal(x, samples, fr)
        assert (cnt == rcnt).all()
        d = np.array([_match_models(cams[h], rcams[h]) for h in range(len(samples)) if cnt[h]])
        # the candidates are as well conditioned as the sample's trifocal tensor; the bulk agrees to round-off
        assert np.mean(d < 1e-8) >= 0.9 and np.median(d) < 1e-10, (np.mean(d < 1e-8), np.median(d))
        if m == 5:
            # property of sfm2d_test.cc:238-272 on exact data: some candidate explains every track at 1e-7
            flat = cams.reshape(-1, 24)
            ok = ~np.isnan(flat).any(axis=1)
            score, inl = fv.score(flat[ok], 1e-7)
            best = np.zeros(len(flat), dtype=int); best[ok] = inl
            assert np.mean(best.reshape(-1, 16).max(axis=1) == 120) >= 0.9
    fv.close()


def test_fourview2d_nonminimal_solver_matches_oracle(oracle):    # sfm2d.cc:446-467
    from privacy_preserving_sfm_amd.device import FourView2dProblem
    sc = synthetic.make_scene_2d(4, 100, n_outliers=20, seed=77)    # sizes of sfm2d_test.cc:238-272
    x = sc["x"] + np.random.default_rng(4).normal(0, 1e-5, sc["x"].shape)
    x = x / np.linalg.norm(x, axis=2, keepdims=True)
    fv = FourView2dProblem(x)
    rng = np.random.default_rng(9)
    samples = np.stack([rng.choice(100, 10, replace=False) for _ in range(256)]).astype(np.int32)
    thr = 1e-3
    cams, score, idx = fv.nonminimal_batch(samples, thr)
    from privacy_preserving_sfm_amd.device import fourview2d_default_frames
    rcams, rcnt = oracle.fourview2d_minimal(x, samples, fourview2d_default_frames())
    agree = 0
    for h in range(len(samples)):
        rs = [oracle.fourview2d_score(rcams[h, k], x, thr)[0] for k in range(rcnt[h])]
        if not rs:
            assert idx[h] == -1
            continue
        # the chosen candidate's score equals the oracle's minimum (candidate order may differ by a sign choice)
        assert abs(score[h] - min(rs)) <= 1e-6 * min(rs) + 1e-9, (h, score[h], min(rs))
        s2 = oracle.fourview2d_score(cams[h], x, thr)[0]
        assert abs(s2 - score[h]) <= 1e-8 * s2 + 1e-11
        agree += 1
    assert agree > 200
    clean = ~sc["is_outlier"][samples].any(axis=1)
    assert clean.any()
    _, inl = fv.score(cams[clean], thr)
    assert inl.max() >= 80
    fv.close()


def test_fourview2d_minimal_full_size_property():
    """4096 samples over 5000 exact tracks: every valid candidate is calibrated (rotation blocks, unit baseline) and
    the best candidate of >= 90 % of the samples explains all tracks."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem
    sc = synthetic.make_scene_2d(4, 5000, seed=8)
    fv = FourView2dProblem(sc["x"])
    rng = np.random.default_rng(0)
    samples = rng.integers(0, 5000, (4096, 5)).astype(np.int32)
    samples = samples[[len(set(s)) == 5 for s in samples]]
    cams, cnt = fv.minimal_batch(samples)
    v = cams[cnt == 16]
    R = v[:, :, 1:3, :, :2]
    RtR = np.einsum("...ki,...kj->...ij", R, R)
    assert np.abs(RtR - np.eye(2)).max() < 1e-5
    assert np.abs(np.linalg.norm(v[:, :, 1, :, 2], axis=-1) - 1).max() < 1e-9
    bc, bs, bi = fv.nonminimal_batch(samples, 1e-7)
    _, inl = fv.score(bc[bi >= 0], 1e-7)
    assert np.mean(inl == 5000) >= 0.9
    fv.close()


def test_pose2d_solver_and_score_match_oracle(oracle):           # sfm2d.cc:491-530, sfm2d_test.cc:112-139
    from privacy_preserving_sfm_amd.device import Pose2dProblem
    sc = synthetic.make_scene_2d(4, 150, n_outliers=30, seed=21)
    rng = np.random.default_rng(2)
    for cam in (1, 3):
        pp = Pose2dProblem(sc["x"][cam] * 2.5, sc["X"])          # the constructor normalises the bearings
        for m in (3, 6, 21):
            samples = np.stack([rng.choice(150, m, replace=False) for _ in range(200)]).astype(np.int32)
            poses = pp.solve_batch(samples)
            for h in range(0, 200, 7):
                ref = oracle.abspose2d_nonminimal(sc["x"][cam], sc["X"], samples[h])
                assert np.abs(poses[h] - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (m, h)
            clean = ~sc["is_outlier"][samples].any(axis=1)
            if clean.any():
                assert np.abs(poses[clean] - sc["cams"][cam]).max() < 1e-8  # exact data: the ground-truth pose
        models = np.concatenate([sc["cams"][cam][None], poses[:40]])
        score, inl = pp.score(models, 1e-6)
        for k in range(len(models)):
            z = sc["X"] @ models[k][:, :2].T + models[k][:, 2]
            e = 1.0 - np.sum(sc["x"][cam] * z / np.linalg.norm(z, axis=1, keepdims=True), axis=1)
            far = np.abs(e - 1e-6) > 1e-12
            assert inl[k] == int(np.sum(e[far] < 1e-6)) + int(np.sum((~far) & (e < 1e-6))) or abs(inl[k] - np.sum(e < 1e-6)) <= np.sum(~far)
            assert abs(score[k] - np.minimum(e, 1e-6).sum()) <= 1e-9 * score[k] + 1e-15
        assert inl[0] == 120
        pp.close()


@pytest.mark.parametrize("n,nout,seed,thr,noise", [(10, 0, 3, 1.0, 0.0), (100, 20, 4, 2e-5, 0.0), (100, 20, 4, 2e-5, 1e-3), (400, 150, 9, 2e-6, 5e-4)])
def test_pose2d_lomsac_matches_oracle(oracle, n, nout, seed, thr, noise):   # sfm2d_test.cc:164-236
    """exact data (the reference's tests): same result; noisy data (scores separated beyond round-off): the same
    trajectory as the restated RansacLib driver — iterations, LO runs, inlier set"""
    from privacy_preserving_sfm_amd.device import Pose2dProblem, lomsac_options
    sc = synthetic.make_scene_2d(4, n, n_outliers=nout, seed=seed)
    rng = np.random.default_rng(seed)
    for cam in (1, 2):
        x = sc["x"][cam] + noise * rng.normal(size=sc["x"][cam].shape)
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
        pp = Pose2dProblem(x, sc["X"])
        rep, pose, idx = pp.lomsac(lomsac_options(squared_inlier_threshold=thr))
        rinl, rP, rst, ridx = oracle.abspose2d_lomsac(x, sc["X"], oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=thr))
        assert rep.best_num_inliers >= n - nout - (0 if noise == 0 else n // 20)
        assert sc["is_outlier"][idx].sum() <= (0 if noise == 0 else 3)        # a random bearing may land inside the threshold
        if noise == 0:
            assert np.linalg.norm(pose - sc["cams"][cam]) < 1e-8 and rep.best_num_inliers == rinl
        else:
            assert np.linalg.norm(pose - sc["cams"][cam]) < 20 * noise
            assert rep.num_iterations == rst.num_iterations and rep.number_lo_iterations == rst.number_lo_iterations
            assert rep.best_num_inliers == rinl and np.array_equal(idx, ridx)
            assert np.abs(pose - rP).max() <= 1e-8
        pp.close()


def test_fourview2d_least_squares_matches_oracle(oracle):          # sfm2d.cc:42-175, 469-489
    """bundle_adjust2d + optimize_points2d (the reference calls Ceres; device and oracle restate the same LM): same
    refined cameras and points, on exact data (fixed point), on noisy data, and with < 10 sample tracks (no bundle)."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem
    sc = synthetic.make_scene_2d(4, 300, seed=11)
    fr = np.zeros(12)
    rng = np.random.default_rng(1)
    for noise, sample in ((0.0, np.arange(0, 300, 9)), (1e-3, np.arange(0, 300, 9)), (1e-3, np.arange(5, 300, 37)), (3e-3, np.arange(300))):
        x = sc["x"] + noise * rng.normal(size=sc["x"].shape)
        x /= np.linalg.norm(x, axis=2, keepdims=True)
        start_cams = sc["cams"].copy()
        if noise > 0:
            start_cams[2][:, 2] += 0.01                       # start away from the minimum
        fv = FourView2dProblem(x)
        cams, X = fv.least_squares(sample, start_cams, sc["X"])
        rcams, rX = oracle.fourview2d_least_squares(x, sample, fr, start_cams, sc["X"])
        assert np.abs(cams - rcams).max() <= 1e-7 * max(1.0, np.abs(rcams).max()), (noise, len(sample), np.abs(cams - rcams).max())
        assert np.abs(X - rX).max() <= 1e-6 * max(1.0, np.abs(rX).max())
        if len(sample) < 10:
            assert np.array_equal(cams, start_cams)
        if noise == 0:
            assert np.abs(cams - sc["cams"]).max() < 1e-9 and np.abs(X - sc["X"]).max() < 1e-8
        fv.close()


@pytest.mark.parametrize("n,nout,seed,thr,noise", [(100, 20, 6, 1e-7, 0.0), (120, 30, 8, 2e-3, 2e-4)])
def test_fourview2d_lomsac(oracle, n, nout, seed, thr, noise):      # sfm2d_test.cc:238-272 (100 tracks, 20 outliers, 1e-7, >= 80 inliers)
    from privacy_preserving_sfm_amd.device import FourView2dProblem, fourview2d_default_frames, lomsac_options
    sc = synthetic.make_scene_2d(4, n, n_outliers=nout, seed=seed)
    rng = np.random.default_rng(seed)
    x = sc["x"] + noise * rng.normal(size=sc["x"].shape)
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    fv = FourView2dProblem(x)
    rep, cams, X, idx = fv.lomsac(lomsac_options(squared_inlier_threshold=thr))
    assert rep.best_num_inliers >= n - nout - (0 if noise == 0 else n // 10)
    assert sc["is_outlier"][idx].sum() <= (0 if noise == 0 else 2)
    # the returned model explains its inliers: errors recomputed from the returned cameras AND points
    err = np.zeros(n)
    for j in range(4):
        z = X @ cams[j][:, :2].T + cams[j][:, 2]
        err = np.maximum(err, np.abs(x[j][:, 0] / x[j][:, 1] - z[:, 0] / z[:, 1]))
    assert (err[idx] < thr).all() and rep.num_inlier_indices == len(idx)
    rinl, rcams, rX, rst, ridx = oracle.fourview2d_lomsac(x, fourview2d_default_frames(), oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=thr))
    assert abs(rep.best_num_inliers - rinl) <= max(1, n // 50)
    if noise > 0:
        # same scene up to the gauge both fix identically (camera 0 = identity, |t_1| = 1): cameras agree
        assert np.abs(cams - rcams).max() < 5e-3
    fv.close()


@pytest.mark.parametrize("seed", [100, 104, 110, 118])
def test_fourview2d_lomsac_takes_the_oracles_trajectory(oracle, seed):
    """Noisy four-view scenes drawn the way tools/fuzz_lomsac.py draws them (round 6: 30 of 30 seeds on the oracle's trajectory, cameras within 2e-10): the
    device-resident LO-MSAC (pooled models, deferred scores, device-side least squares) takes the restated RansacLib driver's trajectory - iterations, local
    optimisations, inlier set - and returns its cameras and points (sfm2d.cc:363-489, sfm2d_test.cc:238-272 shapes)."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem, fourview2d_default_frames, lomsac_options
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30, 400)); nout = int(rng.integers(0, n // 3)); noise = float(10 ** rng.uniform(-4.5, -3.0))
    n = min(n, 160)
    sc = synthetic.make_scene_2d(4, n, n_outliers=min(nout, n // 4), seed=seed)
    x = sc["x"] + noise * rng.normal(size=sc["x"].shape)
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    fv = FourView2dProblem(x)
    rep, cams, X, idx = fv.lomsac(lomsac_options(squared_inlier_threshold=2e-3))
    rinl, rcams, rX, rst, ridx = oracle.fourview2d_lomsac(x, fourview2d_default_frames(), oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=2e-3))
    fv.close()
    assert rep.num_iterations == rst.num_iterations and rep.number_lo_iterations == rst.number_lo_iterations
    assert rep.best_num_inliers == rinl and np.array_equal(idx, ridx)
    # the score sums the errors of tracks whose 50-iteration point refinement has not converged (outlier tracks): 1e-11 in the cameras shows as 1e-7 there
    assert abs(rep.best_model_score - rst.best_model_score) <= 1e-6 * rst.best_model_score
    assert np.abs(cams - rcams).max() <= 1e-7 and np.abs(X[idx] - rX[idx]).max() <= 1e-6 * max(1.0, np.abs(rX[idx]).max())


@pytest.mark.parametrize("seed,views", [(70, (0, 1, 2, 3)), (71, (1, 2, 3)), (72, (0,))])
def test_non_finite_data_follow_the_reference_semantics(oracle, seed, views):
    """A NaN error is never an inlier and makes a model's MSAC score NaN, which no `score < best` accepts: EvaluateModelOnPoint nests
    std::max(e1, std::max(e2, std::max(e3, e4))) (sfm2d.cc:316, initializer.cc:332) and RansacLib scores with std::min(squared_error, threshold)
    (ransac.h:302-305) - both keep a NaN in first position.  With a NaN bearing in the data every model has a NaN error, so the reference accepts none
    (0 inliers after max_num_iterations).  The device's running fmax from 0 and its fmin dropped the NaNs: a NaN MODEL scored 0 on every track and won
    (tools/fuzz_hostile_inputs.py, round 6: "200 inliers" with NaN cameras); the oracle's running std::max had the same fault.  Device = oracle = nothing
    accepted, for the three estimators."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem, Pose2dProblem, PlanarOffsetProblem, fourview2d_default_frames, lomsac_options
    rng = np.random.default_rng(seed)
    sc = synthetic.make_scene_2d(4, 120, n_outliers=20, seed=seed)
    x = sc["x"] + 2e-4 * rng.normal(size=sc["x"].shape)
    for v in views:
        x[v, rng.choice(120, size=3, replace=False), rng.integers(0, 2)] = np.nan
    xn = x / np.linalg.norm(x, axis=2, keepdims=True)
    opt = dict(max_num_iterations=300)
    fv = FourView2dProblem(x)
    rep, cams, X, idx = fv.lomsac(lomsac_options(squared_inlier_threshold=2e-3, **opt))
    fv.close()
    rinl, rcams, rX, rst, ridx = oracle.fourview2d_lomsac(xn, fourview2d_default_frames(), oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=2e-3, **opt))
    assert rep.best_num_inliers == rinl == 0 and rep.num_iterations == rst.num_iterations == 300 and len(idx) == len(ridx) == 0
    xs = sc["x"][1] + 2e-4 * rng.normal(size=sc["x"][1].shape)
    xs[rng.choice(120, size=3, replace=False), 0] = np.nan
    xs = xs / np.linalg.norm(xs, axis=1, keepdims=True)
    pq = Pose2dProblem(xs, sc["X"])
    rep2, pose, idx2 = pq.lomsac(lomsac_options(squared_inlier_threshold=2e-5, **opt))
    pq.close()
    r2, rP, rst2, ridx2 = oracle.abspose2d_lomsac(xs, sc["X"], oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=2e-5, **opt))
    assert rep2.best_num_inliers == r2 == 0 and rep2.num_iterations == rst2.num_iterations == 300
    ps = synthetic.make_planar_offset_scene(120, n_outliers=20, seed=seed, noise=1e-4)
    ls = np.array(ps["lines"])
    ls.reshape(-1)[rng.choice(ls.size, size=4, replace=False)] = np.nan
    ps["lines"] = ls
    pp = PlanarOffsetProblem(ps["poses"], ps["lines"], ps["Rg"])
    rep3, off, cams3, idx3 = pp.lomsac(lomsac_options(squared_inlier_threshold=0.005, **opt))
    pp.close()
    r3, rc3, rst3, ridx3 = oracle.planar_lomsac(ps, oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=0.005, **opt))
    assert rep3.best_num_inliers == r3 == 0 and rep3.num_iterations == rst3.num_iterations == 300


def test_fourview2d_degenerate_samples_never_win(oracle):
    """Clean data in which a third of the tracks are copies of one track: minimal samples that draw two copies are singular and their models come out
    non-finite.  The reference never accepts such a model (its score is NaN); the device and the oracle take the same trajectory and end on a finite
    model that explains the distinct tracks."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem, fourview2d_default_frames, lomsac_options
    sc = synthetic.make_scene_2d(4, 150, n_outliers=0, seed=31)
    rng = np.random.default_rng(31)
    x = sc["x"] + 2e-4 * rng.normal(size=sc["x"].shape)
    x[:, 100:] = x[:, :1]                                     # 50 copies of track 0
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    fv = FourView2dProblem(x)
    rep, cams, X, idx = fv.lomsac(lomsac_options(squared_inlier_threshold=2e-3))
    fv.close()
    rinl, rcams, rX, rst, ridx = oracle.fourview2d_lomsac(x, fourview2d_default_frames(), oracle.LoMsacOptionsC.defaults(squared_inlier_threshold=2e-3))
    assert np.isfinite(cams).all() and np.isfinite(rcams).all()
    assert rep.best_num_inliers == rinl >= 140 and rep.num_iterations == rst.num_iterations and np.array_equal(idx, ridx)
    assert np.abs(cams - rcams).max() <= 1e-6


def test_fourview2d_least_squares_many_points_fallback(oracle):
    """More tracks than the register-resident point refinement takes (kPointsWaveMax = 4096: sixteen workgroups of one point per lane): the many-points
    kernel (`k_fv2d_points`, scale / ratio arrays in memory) refines the same joint problem - cameras and points equal the oracle's (sfm2d.cc:42-175, 469-489)."""
    from privacy_preserving_sfm_amd.device import FourView2dProblem
    sc = synthetic.make_scene_2d(4, 5000, seed=12)
    rng = np.random.default_rng(2)
    x = sc["x"] + 1e-3 * rng.normal(size=sc["x"].shape)
    x /= np.linalg.norm(x, axis=2, keepdims=True)
    start_cams = sc["cams"].copy()
    start_cams[2][:, 2] += 0.01
    sample = np.arange(0, 5000, 97)
    fv = FourView2dProblem(x)
    cams, X = fv.least_squares(sample, start_cams, sc["X"])
    fv.close()
    rcams, rX = oracle.fourview2d_least_squares(x, sample, np.zeros(12), start_cams, sc["X"])
    assert np.abs(cams - rcams).max() <= 1e-7 * max(1.0, np.abs(rcams).max())
    assert np.abs(X - rX).max() <= 1e-6 * max(1.0, np.abs(rX).max())
