"""PPSFM_POOL_POISON=1 python tools/poison_probe.py: a 100-image problem solved and destroyed, then the 500-image headline problem on recycled (poisoned) blocks."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
def step(msg):
    print(msg, flush=True)
small = synthetic.make_ba_scene(100, 5000, 8, seed=0xC0FFEE + 2, model=2)
for mode in sys.argv[1:] or ["eval+solve"]:
    pb = BAProblem(small); step("small created")
    if "eval" in mode: pb.evaluate(); step("small evaluated")
    if "solve" in mode: pb.solve(ba_options(max_num_iterations=3)); step("small solved")
    pb.close(); step("small closed")
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)
    pb = BAProblem(sc); step("big created")
    pb.evaluate(); step("big evaluated")
    s = pb.solve(ba_options(max_num_iterations=3)); step("big solved: cost %.6g, fallbacks %d" % (s.final_cost, s.cholesky_fallbacks))
    pb.close()
