"""Where the headline's milliseconds per LM iteration go OUTSIDE the kernels: per chunk of 10 iterations (bench.run_ba) the wall of set_parameters, of pp_ba_solve
(host wall and HIP-event time of the same call) and the sum of the per-phase events.   gpurun -- python tools/solve_call_overhead_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from privacy_preserving_sfm_amd.device import BAProblem
sc = bench.make_scene(0) if hasattr(bench, "make_scene") else None
if sc is None:
    from privacy_preserving_sfm_amd import synthetic
    sc = synthetic.make_ba_scene(bench.BA_CFG["num_cams"], bench.BA_CFG["num_points"], bench.BA_CFG["track"], seed=0xC0FFEE + 3, model=2)
pb = BAProblem(sc)
bench.run_ba(pb, sc, 10, bench.opts_fn)
for chunk in (10, 20, 50):
    ts, tsol, tdev, ttot = [], [], [], []
    for rep in range(6):
        t0 = time.perf_counter(); pb.set_parameters(sc["poses"], sc["points"], None); t1 = time.perf_counter()
        s = pb.solve(bench.opts_fn(chunk)); t2 = time.perf_counter()
        ts.append(t1 - t0); tsol.append(t2 - t1); tdev.append(s.device_time_s); ttot.append(s.total_time_s)
    med = lambda v: sorted(v)[len(v) // 2]
    print("chunk %2d: set_parameters %.0f us | solve wall %.0f us = %.1f us/it (inside the library %.0f us, HIP events %.0f us = %.1f us/it) | iterations %d successful %d" %
          (chunk, 1e6 * med(ts), 1e6 * med(tsol), 1e6 * med(tsol) / chunk, 1e6 * med(ttot), 1e6 * med(tdev), 1e6 * med(tdev) / chunk, s.num_iterations, s.num_successful_steps))
o = bench.opts_fn(10); o.phase_timings = 1
pb.set_parameters(sc["poses"], sc["points"], None)
pb.solve(o)
t = pb.timings()
print("phases (ms per call, calls):", {k: (round(v[0], 4), v[1]) for k, v in t.items()}, "sum of phase means %.1f us" % (1e3 * sum(v[0] for v in t.values())))
pb.close()
