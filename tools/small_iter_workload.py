"""three 25-iteration solves of a configs[0]-sized problem (20 images / 2000 observations) - the workload behind tools/kernel_gaps.py
   rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -- python tools/small_iter_workload.py; python tools/kernel_gaps.py /tmp/prof_s"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
C, P, t = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (20, 250, 8)
sc = synthetic.make_ba_scene(C, P, t, seed=0xC0FFEE + 1, model=2)
pb = BAProblem(sc)
for r in range(3):
    pb.set_parameters(sc["poses"], sc["points"], None)
    s = pb.solve(ba_options(max_num_iterations=25, gradient_tolerance=0.0))
print("iterations", s.num_iterations)
