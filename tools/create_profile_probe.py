"""The wall of a global bundle-adjustment CALL as the mapper issues it (a new BundleAdjuster per call, src/sfm/incremental_mapper.cc:893-936): pp_ba_create's host
phases (pp_ba_get_create_profile), a 50-iteration solve, read-back, destroy - dense headline scene, banded cfg 3 (ids shuffled), 1000-image sequence.
    gpurun -- python tools/create_profile_probe.py [reps]"""
import sys, time
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scenes = [("dense 500 / 200k", synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)),
          ("banded 500 / 200k, window 40, shuffled", synthetic.shuffle_image_ids(synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, window=40), seed=1)[0]),
          ("banded 1000 / 400k, window 40", synthetic.make_ba_scene(1000, 50000, 8, seed=0xC0FFEE + 3, model=2, window=40))]
for name, sc in scenes:
    rows = []
    for r in range(reps + 1):
        t0 = time.perf_counter(); pb = BAProblem(sc); t1 = time.perf_counter()
        s = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=0.0)); t2 = time.perf_counter()
        pb.get_parameters(); prof = pb.create_profile(); st = pb.structure(); pb.close(); t3 = time.perf_counter()
        if r:
            rows.append((t3 - t0, t1 - t0, t2 - t1, prof))
    rows.sort(key=lambda x: x[0])
    w, c, so, prof = rows[len(rows) // 2]
    print("%-42s wall %6.2f ms | create %5.2f (%.0f %%) = ordering %.2f + pair lists %.2f + structure %.2f + upload %.2f | task plan %.2f | solve %.2f | chains %d steps %d" %
          (name, 1e3 * w, 1e3 * c, 100 * c / w, prof["ordering"], prof["pair_lists"], prof["structure"], prof["upload"], prof["task_plan"], 1e3 * so, st["chains"], st["chain_steps"]), flush=True)
