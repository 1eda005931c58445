"""Repeated solves of one banded problem in its nested-dissection order: LM iterations / s per repetition and the fallback count (a one-launch
factorisation that ran into a bounded wait is repeated with per-column launches).   gpurun -- python tools/nd_stability_probe.py [images] [window] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 40
R = int(sys.argv[3]) if len(sys.argv) > 3 else 12
sc = synthetic.make_ba_scene(C, 50 * C, 8, seed=0xC0FFEE + 3, model=2, window=W)
pb = BAProblem(sc)
print(pb.structure())
rates = []
for r in range(R):
    pb.set_parameters(sc["poses"], sc["points"], None)
    t0 = time.perf_counter(); s = pb.solve(bench.opts_fn(20)); dt = time.perf_counter() - t0
    rates.append(s.num_iterations / dt)
    print("rep %d: %.0f LM it/s, iterations %d, fallbacks %d, solver %d" % (r, rates[-1], s.num_iterations, s.cholesky_fallbacks, s.linear_solver))
pb.close()
