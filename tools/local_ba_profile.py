"""Kernel sequence of one LM iteration of the mapper's local bundle adjustment (6 images, 2004 observations):
    gpurun -- 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lba -- python $GRAFT_REPO_ROOT/tools/local_ba_profile.py'"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
cams = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = synthetic.make_ba_scene(cams, 334 if cams == 6 else 50 * cams, 6 if cams == 6 else 4, seed=0xC0FFEE + 9, model=2)
pb = BAProblem(sc)
o = ba_options(max_num_iterations=25, gradient_tolerance=0.0, function_tolerance=0.0, parameter_tolerance=0.0)
pb.solve(o)
pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
t0 = time.perf_counter(); s = pb.solve(o); dt = time.perf_counter() - t0
print(cams, "images", len(sc["obs_pose"]), "observations:", s.num_iterations, "iterations, %.1f us per LM iteration" % (1e6 * dt / s.num_iterations))
pb.close()
