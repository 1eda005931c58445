"""wall of a whole small BA call (create + solve) and the phase split of its LM iteration - the shape of the mapper's local bundle adjustment
(src/sfm/incremental_mapper.cc:813-858).   gpurun -- python tools/small_ba_probe.py"""
import time, numpy as np, sys
sys.path.insert(0, '.')
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
for (C, P, t) in ((20, 250, 8), (6, 334, 6), (50, 800, 8)):
    sc = synthetic.make_ba_scene(C, P, t, seed=0xC0FFEE + 1, model=2)
    o = ba_options(max_num_iterations=25, gradient_tolerance=0.0)
    for rep in range(3):
        t0 = time.perf_counter(); pb = BAProblem(sc); t1 = time.perf_counter(); s = pb.solve(o); t2 = time.perf_counter()
        pb.set_parameters(sc["poses"], sc["points"], None); t3 = time.perf_counter(); s2 = pb.solve(o); t4 = time.perf_counter()
        if rep == 2:
            pb.set_parameters(sc["poses"], sc["points"], None)
            pb.solve(ba_options(max_num_iterations=25, gradient_tolerance=0.0, phase_timings=1))
            ph = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pb.timings().items()}
        t4b = time.perf_counter(); pb.close(); t5 = time.perf_counter()
    print(C, P, len(sc["obs_pose"]), "create %.2f ms | first solve %.2f ms (%d it) | solve again %.2f ms = %.1f us per iteration | destroy %.2f ms | linsolve %d | phases [us, with event overhead] %s" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, s.num_iterations, (t4 - t3) * 1e3, (t4 - t3) * 1e6 / max(s2.num_iterations, 1), (t5 - t4b) * 1e3, s.linear_solver, ph))
