"""Kernel profile of an LM run with ONE variable camera shared by all images (cfg-3 size; argv[1] = window, 0 = dense):
    gpurun -- 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sintr -- python $GRAFT_REPO_ROOT/tools/shared_intr_profile.py'"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
window = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, num_intrinsics=1, window=window or None)
sc["camera_const_mask"] = np.full(1, 0b0110, dtype=np.uint16)
pb = BAProblem(sc)
print(pb.structure())
o = ba_options(max_num_iterations=50, gradient_tolerance=0.0, function_tolerance=0.0, parameter_tolerance=0.0)
t0 = time.perf_counter(); s = pb.solve(o); dt = time.perf_counter() - t0
print(s.num_iterations, s.final_cost, "%.0f LM it/s" % (s.num_iterations / dt))
pb.close()
