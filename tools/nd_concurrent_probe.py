"""k banded cfg-3 problems (nested-dissection order: four chain workgroups each) solved at once on ONE GPU, a host thread and a stream per handle: aggregate
LM iterations / s and whether any one-launch factorisation ran into a bounded wait (cholesky_fallbacks).   gpurun -- python tools/nd_concurrent_probe.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, window=40)
for k in (1, 2, 4, 8):
    pbs = [BAProblem(sc) for _ in range(k)]
    for pb in pbs:
        bench.run_ba(pb, sc, 5, bench.opts_fn)
    fb = [0] * k
    def work(i):
        pbs[i].set_parameters(sc["poses"], sc["points"], None)
        s = pbs[i].solve(bench.opts_fn(40))
        fb[i] = s.cholesky_fallbacks
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("k=%d handles: %.0f LM it/s aggregate, fallbacks %s, structure chains %d" % (k, 40 * k / dt, fb, pbs[0].structure()["chains"]))
    [pb.close() for pb in pbs]
