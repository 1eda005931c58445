"""LO-MSAC trajectories of the three initialisation problems against the oracle's restated RansacLib driver over many seeds (the GPU suite pins a handful):
planar offsets and 2D absolute pose on noisy data must take the SAME trajectory (iterations, local optimisations, inlier set); the four-view problem is
reported (its minimal solver ends in an eigenvector, so a trajectory may legitimately part ways at a rounding-level tie).
   gpurun -- python tools/fuzz_lomsac.py [seeds] [first seed]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import PlanarOffsetProblem, Pose2dProblem, FourView2dProblem, fourview2d_default_frames, lomsac_options
orc.build()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
same4 = 0
for seed in range(first, first + seeds):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(30, 400)); nout = int(rng.integers(0, n // 3)); noise = float(10 ** rng.uniform(-4.5, -3.0))
    # planar offsets (initializer_test.cc:234-341 shapes)
    sc = synthetic.make_planar_offset_scene(n, n_outliers=nout, seed=seed, noise=noise)
    pp = PlanarOffsetProblem(sc["poses"], sc["lines"], sc["Rg"])
    rep, off, cams, idx = pp.lomsac(lomsac_options(squared_inlier_threshold=0.005))
    inl, rcams, st, ridx = orc.planar_lomsac(sc, orc.LoMsacOptionsC.defaults(squared_inlier_threshold=0.005))
    pp.close()
    ok_p = (rep.best_num_inliers == inl and np.array_equal(idx, ridx) and rep.num_iterations == st.num_iterations and
            rep.number_lo_iterations == st.number_lo_iterations and np.allclose(cams, rcams, rtol=1e-7, atol=1e-9))
    # 2D absolute pose (sfm2d_test.cc:164-236 shapes)
    s2 = synthetic.make_scene_2d(4, n, n_outliers=nout, seed=seed)
    thr = float(10 ** rng.uniform(-5.5, -4.5))
    x = s2["x"][1] + noise * rng.normal(size=s2["x"][1].shape); x = x / np.linalg.norm(x, axis=1, keepdims=True)
    pq = Pose2dProblem(x, s2["X"])
    rep2, pose, idx2 = pq.lomsac(lomsac_options(squared_inlier_threshold=thr))
    rinl, rP, rst, ridx2 = orc.abspose2d_lomsac(x, s2["X"], orc.LoMsacOptionsC.defaults(squared_inlier_threshold=thr))
    pq.close()
    ok_a = (rep2.num_iterations == rst.num_iterations and rep2.number_lo_iterations == rst.number_lo_iterations and rep2.best_num_inliers == rinl and
            np.array_equal(idx2, ridx2) and np.abs(pose - rP).max() <= 1e-8)
    # four views (sfm2d_test.cc:238-272 shapes)
    n4 = min(n, 160); s4 = synthetic.make_scene_2d(4, n4, n_outliers=min(nout, n4 // 4), seed=seed)
    x4 = s4["x"] + noise * rng.normal(size=s4["x"].shape); x4 /= np.linalg.norm(x4, axis=2, keepdims=True)
    fv = FourView2dProblem(x4)
    rep4, cams4, X4, idx4 = fv.lomsac(lomsac_options(squared_inlier_threshold=2e-3))
    rinl4, rcams4, rX4, rst4, ridx4 = orc.fourview2d_lomsac(x4, fourview2d_default_frames(), orc.LoMsacOptionsC.defaults(squared_inlier_threshold=2e-3))
    fv.close()
    same = (rep4.num_iterations == rst4.num_iterations and rep4.number_lo_iterations == rst4.number_lo_iterations and np.array_equal(idx4, ridx4))
    same4 += same
    d4 = np.abs(cams4 - rcams4).max()
    bad += (not ok_p) + (not ok_a)
    print("seed %3d n %3d out %3d noise %.1e | planar %s (%d it, %d inl) | pose2d %s (%d it, %d inl) | four views: %s, inliers %d / %d, cameras differ %.1e" %
          (seed, n, nout, noise, "same" if ok_p else "DIFFERENT", rep.num_iterations, inl, "same" if ok_a else "DIFFERENT", rep2.num_iterations, rinl,
           "same trajectory" if same else "another trajectory (%d/%d it, %d/%d LO)" % (rep4.num_iterations, rst4.num_iterations, rep4.number_lo_iterations, rst4.number_lo_iterations),
           rep4.best_num_inliers, rinl4, d4), flush=True)
print("%d of %d planar / pose2d runs left the oracle's trajectory; four views: %d of %d on the oracle's trajectory" % (bad, 2 * seeds, same4, seeds))
