"""K1 (line residual + Jacobians) against the oracle's jets on HOSTILE geometry, every camera model: points behind cameras, at 1e-7 of the image plane, far away,
quaternions far from unit length (ambient Jacobian), strong distortion.  Asserted per entry: the same finite / non-finite pattern and agreement where finite (relative to the
row's largest entry).   gpurun -- python tools/fuzz_line_eval.py [seeds] [first seed]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
orc.build()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
first = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst = 0.0; worst_near = 0.0; bad = 0
for seed in range(first, first + seeds):
    for model in range(11):
        rng = np.random.default_rng(1000 * seed + model)
        sc = synthetic.make_ba_scene(10, 400, 4, seed=seed * 31 + model, model=model, num_intrinsics=2, sort="pose")
        P = len(sc["points"])
        kind = rng.integers(0, 6, size=P)
        pts = np.array(sc["points"])
        pts[kind == 1] *= 8.0                                             # beyond the camera circle: behind some of its cameras
        pts[kind == 2] *= 1e5                                             # far away
        centre_like = np.array([4.0, 0.0, 0.0]) + 1e-7 * rng.normal(size=(int((kind == 3).sum()), 3))
        pts[kind == 3] = centre_like                                      # next to the first camera's centre: depth ~ 1e-7
        sc["points"] = pts
        unit_poses = np.array(sc["poses"])
        unit_poses[:, :4] /= np.linalg.norm(unit_poses[:, :4], axis=1, keepdims=True)
        long_poses = unit_poses.copy()
        long_poses[:, :4] *= 10 ** rng.uniform(-1.5, 1.5, size=(len(long_poses), 1))   # quaternions of length 0.03 .. 30: the AMBIENT Jacobian only (the tangent
        #                                                                               one assumes unit length, as the reference's cost function does; header)
        intr = np.array(sc["intr"])
        if intr.shape[1] > 4:
            intr[:, 4:] *= rng.uniform(-6, 6, size=intr[:, 4:].shape)     # distortion parameters up to 6 x the defaults, either sign
        sc["intr"] = intr
        for ambient in (False, True):
            sc["poses"] = long_poses if ambient else unit_poses
            sc["pose_const"] = np.full(len(unit_poses), 1 if ambient else 0, np.uint8)      # (pp_ba_set_parameters refuses a VARIABLE pose that is not of unit length)
            pb = BAProblem(sc)
            cost, r, jp, jx, jc = pb.evaluate(ambient=ambient, want_cam=True)
            pb.close()
            r0, jp0, jx0, jc0 = orc.ba_eval(sc, ambient=ambient, want_cam=True)
            near = (kind[np.asarray(sc["obs_point"])] == 3) & (not ambient)      # depth ~ 1e-7 behind a cancellation of two numbers of size 4: conditioning, reported apart
            for name, a, b in (("r", r, r0), ("Jpose", jp, jp0), ("Jpoint", jx, jx0), ("Jcam", jc, jc0)):
                a = np.asarray(a); b = np.asarray(b)
                fa, fb = np.isfinite(a), np.isfinite(b)
                if not np.array_equal(fa, fb):
                    bad += 1; print("seed %d model %d %s ambient %d: finite pattern differs at %d entries" % (seed, model, name, ambient, int((fa != fb).sum())), flush=True)
                both = fa & fb
                a2 = np.where(both, a, 0.0).reshape(len(sc["lines"]), -1); b2 = np.where(both, b, 0.0).reshape(len(sc["lines"]), -1)
                scale = np.maximum(np.abs(b2).max(axis=1, keepdims=True), 1.0)
                e = (np.abs(a2 - b2) / scale).max(axis=1)
                worst = max(worst, e[~near].max()); worst_near = max(worst_near, e[near].max() if near.any() else 0.0)
                if e[~near].max() > 1e-8:
                    bad += 1; i = int(np.argmax(np.where(near, 0.0, e))); j = int(np.argmax(np.abs(a2[i] - b2[i])))
                    print("seed %d model %d %s ambient %d: %.2e at observation %d (point class %d; device %.17g oracle %.17g, row scale %.3g)" %
                          (seed, model, name, ambient, e[i], i, kind[sc["obs_point"][i]], a2[i, j], b2[i, j], scale[i, 0]), flush=True)
    print("seed %d done, worst row-relative difference so far %.2e" % (seed, worst), flush=True)
print("%d findings; worst row-relative difference %.2e; on the points at depth 1e-7 (cancellation in R X + t): %.2e" % (bad, worst, worst_near))
