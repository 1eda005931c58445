"""K1 (k_line_eval) rate against the launch size: GB/s of the 220 B/obs algorithmic traffic at M = 200k (BASELINE cfg-3),
1M and 2M observations — how much of the cfg-3 figure is launch ramp / tail."""
import sys
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
for cams, pts in ((500, 25000), (500, 125000), (500, 250000)):
    sc = synthetic.make_ba_scene(cams, pts, 8, seed=1, model=2)
    pb = BAProblem(sc)
    pb.evaluate_device(repeat=20)
    ms = pb.evaluate_device(repeat=100)
    M = pb.M
    print("M = %8d: %.2f us per launch, %.0f GB/s (%.3f of 8 TB/s)" % (M, ms * 1e3, 220.0 * M / (ms * 1e-3) / 1e9, 220.0 * M / (ms * 1e-3) / 8e12))
    pb.close()
