"""A/B of the chunk kernel's processing order (PPSFM_BA_CHUNK_XCD=0: pair order as before; default: by column image in eight XCD runs) on the banded scenes:
LM iterations / s, Schur phase per call, end points (the sums are the same: bitwise equal parameters).   gpurun -- python tools/chunk_xcd_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
VAR = sys.argv[1] if len(sys.argv) > 1 else "PPSFM_BA_CHUNK_XCD"      # the switch to A/B (0 / 1 alternately)
MODES = ("0", "1", "0", "1")
for C, W, shuffle in ((500, 40, 0), (500, 40, 1), (1000, 40, 0), (300, 20, 0)):
    sc = synthetic.make_ba_scene(C, 50 * C, 8, seed=0xC0FFEE + 3, model=2, window=W)
    if shuffle:
        sc, _ = synthetic.shuffle_image_ids(sc, seed=1)
    BAProblem(sc).close()
    end = {}
    for mode in MODES:
        os.environ[VAR] = mode
        pb = BAProblem(sc)
        bench.run_ba(pb, sc, 10, bench.opts_fn)
        rates = []
        for r in range(3):
            t0 = time.perf_counter(); bench.run_ba(pb, sc, 20, bench.opts_fn); rates.append(20 / (time.perf_counter() - t0))
        o = bench.opts_fn(20); o.phase_timings = 1
        pb.set_parameters(sc["poses"], sc["points"], None)
        s = pb.solve(o)
        t = pb.timings()
        print("%4d images window %d%s  switch %s: %s LM it/s, schur %.1f us, cholesky %.1f us per call" %
              (C, W, " shuffled" if shuffle else "", mode, " ".join("%.0f" % v for v in rates), 1e3 * t["schur"][0], 1e3 * t["cholesky"][0]), flush=True)
        end[mode] = pb.get_parameters()
        pb.close()
    print("      parameters bitwise equal between the two orders:", all(np.array_equal(a, b) for a, b in zip(end["0"][:2], end["1"][:2])))
os.environ.pop(VAR)
