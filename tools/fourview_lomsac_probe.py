"""The four-view LO-MSAC leg of bench.py on its own (2000 tracks, 400 outliers, 4096 iterations): wall, LO runs, device time of the minimal solves.
gpurun -- python tools/fourview_lomsac_probe.py [repeats]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import FourView2dProblem, lomsac_options
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
isc = synthetic.make_scene_2d(4, 2000, n_outliers=400, seed=7)
fv = FourView2dProblem(isc["x"], device=0)
fv.lomsac(lomsac_options(squared_inlier_threshold=1e-6, min_num_iterations=256, max_num_iterations=256))
for r in range(reps):
    t0 = time.perf_counter()
    rep, cams, X, inl = fv.lomsac(lomsac_options(squared_inlier_threshold=1e-6, min_num_iterations=4096, max_num_iterations=4096))
    dt = time.perf_counter() - t0
    print("run %d: wall %.1f ms  iterations %d  LO runs %d  minimal+score on the device %.2f ms  inliers %d  score %.12e" %
          (r, 1e3 * dt, rep.num_iterations, rep.number_lo_iterations, 1e3 * rep.device_time_s, rep.best_num_inliers, rep.best_model_score), flush=True)
fv.close()
