"""K8 parity: batched robust track triangulation (one LORANSAC per track, reference estimators/triangulation.cc:55-149,
optim/loransac.h:88-235) against the CPU restatement.  The RANSAC draws no random numbers (CombinationSampler), so
success flags, trial counts and inlier masks are compared exactly on tracks whose residuals are separated from the
threshold; points to 1e-9 (minimal-sample null vector by 3x3 minors vs Jacobi on the Gram matrix)."""
import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("residual_type,max_error,min_angle", [(0, 2e-3, 0.0), (1, 2.0, 0.02), (0, 2e-3, 0.2)])
def test_triangulate_tracks_matches_oracle(oracle, residual_type, max_error, min_angle):
    from privacy_preserving_sfm_amd.device import triangulate_tracks, triangulation_options
    sc = synthetic.make_track_scene(14, 1500, seed=3 + residual_type)
    opt = triangulation_options(min_tri_angle=min_angle, residual_type=residual_type, max_error=max_error, confidence=0.9999, min_inlier_ratio=0.02)
    ok, xyz, mask, nt, ms = triangulate_tracks(sc["track_start"], sc["lines"], sc["obs_view"], sc["P"], sc["centers"], sc["view_camera"], sc["camera_model"],
                                               sc["intr"], sc["cam_size"], opt)
    rok, rxyz, rmask, rnt = oracle.triangulate_tracks(sc, min_angle, residual_type, max_error=max_error, confidence=0.9999, min_inlier_ratio=0.02)
    agree = (ok == rok)
    assert agree.mean() >= 0.995, agree.mean()
    both = ok & rok
    same_trials = (nt == rnt)[both].mean()
    assert same_trials >= 0.99, same_trials
    ts = sc["track_start"]
    # a support of exactly three observations is an EXACT fit (three planes meet in a point): every such sample scores a
    # residual sum of rounding noise and the winner among them is arbitrary in any implementation — compare the tracks
    # whose winner is decided by real residuals (>= 4 inliers)
    ninl = np.array([mask[ts[t]:ts[t + 1]].sum() for t in range(len(ts) - 1)])
    rinl = np.array([rmask[ts[t]:ts[t + 1]].sum() for t in range(len(ts) - 1)])
    decided = np.nonzero(both & (ninl >= 4) & (rinl >= 4))[0]
    assert len(decided) > 800
    same_mask = np.array([np.array_equal(mask[ts[t]:ts[t + 1]], rmask[ts[t]:ts[t + 1]]) for t in decided])
    assert same_mask.mean() >= 0.999, same_mask.mean()
    assert (ninl[both] == rinl[both]).mean() >= 0.995                # the SUPPORT (inlier count) agrees also on the exact fits
    good = decided[same_mask]
    err = np.abs(xyz[good] - rxyz[good]).max(axis=1)
    assert np.mean(err < 1e-8) >= 0.99 and np.median(err) < 1e-11
    # and they are the right points: tracks with >= 4 clean observations triangulate to the truth
    clean = np.array([(~sc["is_outlier"][ts[t]:ts[t + 1]]).sum() for t in range(len(ts) - 1)])
    sel = ok & (clean >= 5)
    assert sel.sum() > 300
    assert np.median(np.linalg.norm(xyz[sel] - sc["points"][sel], axis=1)) < 5e-3
    if min_angle == 0.0:
        assert ok[clean >= 5].mean() > 0.95
    # a failed track reports an empty mask
    for t in np.nonzero(~ok)[0][:50]:
        assert not mask[ts[t]:ts[t + 1]].any()


def test_triangulate_tracks_edge_cases(oracle):
    from privacy_preserving_sfm_amd.device import triangulate_tracks, triangulation_options
    sc = synthetic.make_track_scene(8, 40, seed=9, min_len=2, max_len=4, outlier_frac=0.0, noise=0.0)
    opt = triangulation_options(min_tri_angle=0.0, residual_type=1, max_error=1.0)
    ok, xyz, mask, nt, ms = triangulate_tracks(sc["track_start"], sc["lines"], sc["obs_view"], sc["P"], sc["centers"], sc["view_camera"], sc["camera_model"],
                                               sc["intr"], sc["cam_size"], opt)
    lens = np.diff(sc["track_start"])
    assert not ok[lens < 3].any()                                   # "if(point_data.size() < 3) return false" (triangulation.cc:124-125)
    assert ok[lens >= 3].all() and np.abs(xyz[lens >= 3] - sc["points"][lens >= 3]).max() < 1e-8
    assert (nt[lens == 3] == 1).all()                               # C(3,3) = 1 combination
    # a tiny image rejects every projection: no track survives
    sc2 = dict(sc, cam_size=np.array([[1, 1]], dtype=np.int32))
    ok2, *_ = triangulate_tracks(sc2["track_start"], sc2["lines"], sc2["obs_view"], sc2["P"], sc2["centers"], sc2["view_camera"], sc2["camera_model"],
                                 sc2["intr"], sc2["cam_size"], opt)
    assert not ok2.any()
    # an impossible triangulation angle rejects every sample
    opt3 = triangulation_options(min_tri_angle=1.6, residual_type=1, max_error=1.0)
    ok3, *_ = triangulate_tracks(sc["track_start"], sc["lines"], sc["obs_view"], sc["P"], sc["centers"], sc["view_camera"], sc["camera_model"], sc["intr"],
                                 sc["cam_size"], opt3)
    assert not ok3.any()


def test_triangulate_tracks_full_size():
    """200k observations in 25k tracks (the BA benchmark's shape): clean long tracks come back at the truth"""
    from privacy_preserving_sfm_amd.device import triangulate_tracks, triangulation_options
    sc = synthetic.make_track_scene(64, 25000, seed=1, min_len=8, max_len=8, outlier_frac=0.1)
    opt = triangulation_options(min_tri_angle=0.02, residual_type=0, max_error=2e-3)
    ok, xyz, mask, nt, ms = triangulate_tracks(sc["track_start"], sc["lines"], sc["obs_view"], sc["P"], sc["centers"], sc["view_camera"], sc["camera_model"],
                                               sc["intr"], sc["cam_size"], opt)
    assert ok.mean() > 0.97
    assert np.median(np.linalg.norm(xyz[ok] - sc["points"][ok], axis=1)) < 2e-3
    assert mask[~sc["is_outlier"]].mean() > 0.9 and mask[sc["is_outlier"]].mean() < 0.1


def test_triangulate_long_tracks(oracle):
    """Tracks longer than a wavefront (65-150 observations of one point): the same trials, inlier masks and points as the CPU restatement
    (estimators/triangulation.cc:55-149; the exhaustive CombinationSampler is cut short by the dynamic trial bound, optim/loransac.h:88-235)."""
    from privacy_preserving_sfm_amd.device import triangulate_tracks, triangulation_options
    sc = synthetic.make_track_scene(150, 30, seed=150, min_len=65, max_len=150)
    ts = sc["track_start"]
    assert np.diff(ts).max() > 64
    for residual_type, max_error in ((0, 2e-3), (1, 2.0)):
        opt = triangulation_options(min_tri_angle=0.02, residual_type=residual_type, max_error=max_error, confidence=0.9999, min_inlier_ratio=0.02)
        ok, xyz, mask, nt, ms = triangulate_tracks(ts, sc["lines"], sc["obs_view"], sc["P"], sc["centers"], sc["view_camera"], sc["camera_model"], sc["intr"], sc["cam_size"], opt)
        rok, rxyz, rmask, rnt = oracle.triangulate_tracks(sc, 0.02, residual_type, max_error=max_error, confidence=0.9999, min_inlier_ratio=0.02)
        assert np.array_equal(ok, rok) and ok.all() and np.array_equal(nt, rnt) and np.array_equal(mask, rmask)
        assert np.abs(xyz - rxyz).max() < 1e-10
