"""Non-finite and degenerate values through every estimator of the C ABI: NaN / Inf / zeros / 1e300 written into random places of otherwise valid inputs.
The contract checked: the call RETURNS (an error code or a result - never a hang: every device wait is bounded) within its time limit, and the process
survives.  One child process per case, each under its own timeout, so that a hang would cost one case and not the box.
   gpurun -- python tools/fuzz_hostile_inputs.py"""
import os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd._capi import PPError
from privacy_preserving_sfm_amd import device as D
what, poison, seed = sys.argv[1], sys.argv[2], int(sys.argv[3])
rng = np.random.default_rng(seed)
val = {"nan": np.nan, "inf": np.inf, "ninf": -np.inf, "zero": 0.0, "huge": 1e300, "allones": np.frombuffer(b"\xff" * 8, dtype=np.float64)[0]}[poison]
def hit(a, frac=0.02):
    a = np.array(a, dtype=np.float64); flat = a.reshape(-1)
    idx = rng.choice(flat.size, size=max(1, int(frac * flat.size)), replace=False); flat[idx] = val
    return a
out = "returned"
try:
    if what == "ba_points":
        sc = synthetic.make_ba_scene(30, 1500, 5, seed=seed); sc["points"] = hit(sc["points"])
        pb = D.BAProblem(sc); s = pb.solve(D.ba_options(max_num_iterations=6)); out = "termination %%d after %%d iterations" %% (s.termination, s.num_iterations); pb.close()
    elif what == "ba_tvec":
        sc = synthetic.make_ba_scene(30, 1500, 5, seed=seed); p = np.array(sc["poses"]); p[:, 4:] = hit(p[:, 4:]); sc["poses"] = p
        pb = D.BAProblem(sc); s = pb.solve(D.ba_options(max_num_iterations=6)); out = "termination %%d after %%d iterations" %% (s.termination, s.num_iterations); pb.close()
    elif what == "ba_intr":
        sc = synthetic.make_ba_scene(30, 1500, 5, seed=seed); sc["intr"] = hit(sc["intr"], 0.2)
        pb = D.BAProblem(sc); s = pb.solve(D.ba_options(max_num_iterations=6)); out = "termination %%d after %%d iterations" %% (s.termination, s.num_iterations); pb.close()
    elif what == "ba_iterative":
        sc = synthetic.make_ba_scene(40, 2000, 5, seed=seed); sc["points"] = hit(sc["points"])
        pb = D.BAProblem(sc, linear_solver=2); s = pb.solve(D.ba_options(max_num_iterations=6)); out = "termination %%d after %%d iterations" %% (s.termination, s.num_iterations); pb.close()
    elif what == "ransac":
        sc = synthetic.make_ransac_scene(600, outlier_ratio=0.3, noise_px=0.3, seed=seed, aligned_ratio=0.2)
        pp = D.PoseProblem(hit(sc["lines"]), hit(sc["points"]), sc["aligned"])
        rep, mask = pp.ransac(D.ransac_options(max_error=sc["max_error"], seed=1, min_num_trials=100, max_num_trials=2000)); out = "success %%d trials %%d inliers %%d" %% (rep.success, rep.num_trials, rep.num_inliers); pp.close()
    elif what == "triangulation":
        sc = synthetic.make_track_scene(10, 800, seed=seed)
        opt = D.triangulation_options(min_tri_angle=0.0, residual_type=0, max_error=2e-3, confidence=0.9999, min_inlier_ratio=0.02)
        ok, xyz, mask, nt, ms = D.triangulate_tracks(sc["track_start"], hit(sc["lines"]), sc["obs_view"], hit(sc["P"]), hit(sc["centers"]), sc["view_camera"], sc["camera_model"], sc["intr"], sc["cam_size"], opt)
        out = "%%d of %%d tracks triangulated" %% (int(ok.sum()), len(ok))
    elif what == "fourview":
        sc = synthetic.make_scene_2d(4, 200, n_outliers=40, seed=seed)
        fv = D.FourView2dProblem(hit(sc["x"])); rep, cams, X, idx = fv.lomsac(D.lomsac_options(squared_inlier_threshold=2e-3)); out = "inliers %%d after %%d iterations" %% (rep.best_num_inliers, rep.num_iterations); fv.close()
    elif what == "pose2d":
        sc = synthetic.make_scene_2d(4, 200, n_outliers=40, seed=seed)
        pq = D.Pose2dProblem(hit(sc["x"][1]), hit(sc["X"])); rep, pose, idx = pq.lomsac(D.lomsac_options(squared_inlier_threshold=2e-5)); out = "inliers %%d after %%d iterations" %% (rep.best_num_inliers, rep.num_iterations); pq.close()
    elif what == "planar":
        sc = synthetic.make_planar_offset_scene(200, n_outliers=40, seed=seed, noise=1e-4)
        pp = D.PlanarOffsetProblem(sc["poses"], hit(sc["lines"]), sc["Rg"]); rep, off, cams, idx = pp.lomsac(D.lomsac_options(squared_inlier_threshold=0.005)); out = "inliers %%d after %%d iterations" %% (rep.best_num_inliers, rep.num_iterations); pp.close()
except PPError as e:
    out = "refused: code %%d (%%s)" %% (e.code, str(e)[:90])
print("RESULT " + out)
''' % ROOT
limit = 90
bad = 0
cases = [(w, p) for w in ("ba_points", "ba_tvec", "ba_intr", "ba_iterative", "ransac", "triangulation", "fourview", "pose2d", "planar") for p in ("nan", "inf", "zero", "huge", "allones")]
for i, (what, poison) in enumerate(cases):
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, what, poison, str(40 + i)], capture_output=True, text=True, timeout=limit)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        status = line[0][7:] if line else "NO RESULT (exit %d): %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1:] )
        if not line: bad += 1
    except subprocess.TimeoutExpired:
        status = "TIMEOUT after %d s" % limit; bad += 1
    print("%-14s %-8s %5.1f s  %s" % (what, poison, time.time() - t0, status), flush=True)
    if "TIMEOUT" in status:
        print("stopping at the first hang"); break
print("%d of %d cases hung or died" % (bad, len(cases)))
