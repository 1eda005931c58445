"""the DIRECT solve with variable intrinsics at the headline size (500 images / 200k observations; one shared SIMPLE_RADIAL camera, then a camera per
image; f and k variable): LM iterations / s and the phase split.   gpurun -- python tools/intr_direct_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
for nintr in ([int(a) for a in sys.argv[1:]] or (1, 500)):
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, num_intrinsics=nintr)
    sc["camera_const_mask"] = np.full(nintr, 0b0110, dtype=np.uint16)
    t0 = time.perf_counter(); pb = BAProblem(sc); t1 = time.perf_counter()
    pb.solve(ba_options(max_num_iterations=3))
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    o = ba_options(max_num_iterations=10, phase_timings=1)
    t2 = time.perf_counter(); s = pb.solve(o); t3 = time.perf_counter()
    print(nintr, "create %.0f ms, %.0f LM it/s, linear solver %d" % ((t1 - t0) * 1e3, s.num_iterations / (t3 - t2), s.linear_solver), {k: round(v[0] / max(v[1], 1) * 1e3) for k, v in pb.timings().items()}, flush=True)
    pb.close()
