"""ITERATIVE_SCHUR with variable intrinsics at 1100 images (one shared SIMPLE_RADIAL camera, f and k variable) against the direct solve of the same
problem (6600 + 2 columns).   gpurun -- python tools/pcg_intr_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
only = [int(a) for a in sys.argv[1:]]
for nintr in (only or (1, 1100)):
    sc = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2, num_intrinsics=nintr)
    sc["camera_const_mask"] = np.full(nintr, 0b0110, dtype=np.uint16)
    for ls, name in ((0, "auto (iterative)"),) + ((() if only else ((1, "direct"),))):
        t0 = time.perf_counter(); pb = BAProblem(sc, linear_solver=ls); t1 = time.perf_counter()
        pb.solve(ba_options(max_num_iterations=3))
        pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
        t2 = time.perf_counter(); s = pb.solve(ba_options(max_num_iterations=10)); t3 = time.perf_counter()
        print("%d intrinsics blocks, %s: create %.0f ms, %.0f LM it/s, cost %.3e -> %.3e, linear solver %d, cg iterations %d" %
              (nintr, name, (t1 - t0) * 1e3, s.num_iterations / (t3 - t2), s.initial_cost, s.final_cost, s.linear_solver, s.linear_solver_iterations), flush=True)
        pb.close()
