"""LM iterations / s of the iterative path (ITERATIVE_SCHUR + SCHUR_JACOBI) at 1100 images.   gpurun -- python tools/pcg_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
sc = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2)
pb = BAProblem(sc)
bench.run_ba(pb, sc, 10, bench.opts_fn)
for r in range(3):
    t0 = time.perf_counter(); bench.run_ba(pb, sc, 20, bench.opts_fn); dt = time.perf_counter() - t0
    print("pcg 1100: %.0f LM it/s" % (20 / dt))
pb.set_parameters(sc["poses"], sc["points"], None)
s = pb.solve(bench.opts_fn(10)); print("cg its", s.linear_solver_iterations, "cost", s.final_cost)
o = bench.opts_fn(20); o.phase_timings = 1
pb.set_parameters(sc["poses"], sc["points"], None)
s = pb.solve(o)
print("phases (ms per call, calls):", {k: (round(v[0], 4), v[1]) for k, v in pb.timings().items()}, "cg its", s.linear_solver_iterations)
