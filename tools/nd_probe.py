"""The banded cfg-3 scene (500 images, every point inside a 40-image window) with its camera order as pp_ba_create chooses it (nested dissection: several
chains) against the band order alone (PPSFM_BA_ORDERING=band: one chain): structure, LM iterations / s, per-phase timings, and the two solves' end points.
   gpurun -- python tools/nd_probe.py [images] [window] [shuffle] [loop]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
C = int(sys.argv[1]) if len(sys.argv) > 1 else 500
W = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sc = synthetic.make_ba_scene(C, 50 * C, 8, seed=0xC0FFEE + 3, model=2, window=W, loop=len(sys.argv) > 4 and bool(int(sys.argv[4])))
if len(sys.argv) > 3 and int(sys.argv[3]):
    sc, _ = synthetic.shuffle_image_ids(sc, seed=1)
end = {}
BAProblem(sc).close()      # (the first handle of a process pays the first-time allocations: not part of the comparison)
for mode in ("", "band"):
    if mode:
        os.environ["PPSFM_BA_ORDERING"] = mode
    else:
        os.environ.pop("PPSFM_BA_ORDERING", None)
    t0 = time.perf_counter(); pb = BAProblem(sc); tc = time.perf_counter() - t0
    print("ordering=%s: create %.1f ms, structure %s" % (mode or "auto", 1e3 * tc, pb.structure()))
    bench.run_ba(pb, sc, 10, bench.opts_fn)
    for r in range(3):
        t0 = time.perf_counter(); bench.run_ba(pb, sc, 20, bench.opts_fn); dt = time.perf_counter() - t0
        print("   %.0f LM it/s" % (20 / dt))
    o = bench.opts_fn(20); o.phase_timings = 1
    pb.set_parameters(sc["poses"], sc["points"], None)
    s = pb.solve(o)
    print("   phases (ms per call, calls):", {k: (round(v[0], 4), v[1]) for k, v in pb.timings().items()}, "cost", s.final_cost, "fallbacks", getattr(s, "cholesky_fallbacks", None))
    end[mode] = (pb.get_parameters()[0].copy(), s.final_cost)
    pb.close()
d = np.abs(end[""][0] - end["band"][0]).max()
print("end points: max |pose difference| %.3e, costs %.15e / %.15e" % (d, end[""][1], end["band"][1]))
