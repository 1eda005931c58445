"""wall of a whole small BA call (create + solve) - the shape of the mapper's local bundle adjustment (src/sfm/incremental_mapper.cc:813-858)"""
import time, numpy as np, sys
sys.path.insert(0, '.')
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
for (C, P, t) in ((20, 250, 8), (6, 334, 6), (50, 800, 8)):
    sc = synthetic.make_ba_scene(C, P, t, seed=0xC0FFEE + 1, model=2)
    o = ba_options(max_num_iterations=25, gradient_tolerance=0.0)
    for rep in range(4):
        t0 = time.perf_counter(); pb = BAProblem(sc); t1 = time.perf_counter(); s = pb.solve(o); t2 = time.perf_counter(); s2 = pb.solve(o); t3 = time.perf_counter()
        pb.close(); t4 = time.perf_counter()
        print(C, P, len(sc["obs_pose"]), "rep", rep, "create %.2f ms | first solve %.2f ms (%d it) | second solve %.2f ms (%d it) | destroy %.2f ms | linsolve %d" %
              ((t1 - t0) * 1e3, (t2 - t1) * 1e3, s.num_iterations, (t3 - t2) * 1e3, s2.num_iterations, (t4 - t3) * 1e3, s.linear_solver))
