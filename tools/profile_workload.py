#!/usr/bin/env python3
"""Small fixed workload for rocprofv3: the same kernels as bench.py at the same sizes, few launches.
   python tools/profile_workload.py [k1|k1big|ba|band|pcg|ransac|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, PoseProblem, ba_options

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("k1", "ba", "all"):
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)
    pb = BAProblem(sc)
    if what in ("k1", "all"):
        ms = pb.evaluate_device(repeat=50)
        print("k1 ms/launch", ms)
    if what in ("ba", "all"):
        s = pb.solve(ba_options(max_num_iterations=3))
        print("ba iterations", s.num_iterations, "device_s", s.device_time_s, pb.timings())
    pb.close()
if what == "k1big":      # 2M observations: 440 MB per K1 launch, beyond the 256 MiB Infinity Cache
    sc = synthetic.make_ba_scene(500, 250000, 8, seed=1, model=2)
    pb = BAProblem(sc)
    print("k1big ms/launch", pb.evaluate_device(repeat=10))
    pb.close()
if what == "band":      # (not part of `all`: its k_cholesky_tasks launches must not be averaged with the dense headline's)  cfg-3 size, every point inside a 40-image window: block-sparse system, images ordered by nested dissection, four chain workgroups
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, window=40)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=5))
    print("band: structure", pb.structure(), "LM iterations", s.num_iterations, "device_s", s.device_time_s, "fallbacks", s.cholesky_fallbacks)
    pb.close()
if what in ("pcg", "all"):      # 1100 images: the handle picks ITERATIVE_SCHUR + SCHUR_JACOBI by the image count (matrix-free PCG, ba_pcg.hip)
    sc = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=5))
    print("pcg: LM iterations", s.num_iterations, "cg iterations", s.linear_solver_iterations, "device_s", s.device_time_s)
    pb.close()
if what in ("ransac", "all"):
    rsc = synthetic.make_ransac_scene(50000, outlier_ratio=0.5, noise_px=0.5, seed=0xBADC0DE)
    pp = PoseProblem(rsc["lines"], rsc["points"], rsc["aligned"])
    rep = pp.hypotheses(16384, rsc["max_error"] ** 2, seed=0)
    print("ransac hyp/s", 16384 / rep.device_time_s, "models", rep.models_scored)
    pp.close()
