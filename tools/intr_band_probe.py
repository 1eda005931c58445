import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
for nintr in (1, 500):
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, num_intrinsics=nintr, window=40)
    sc["camera_const_mask"] = np.full(nintr, 0b0110, dtype=np.uint16)
    BAProblem(sc).close()
    for mode in ("", "0"):
        if mode: os.environ["PPSFM_BA_SPARSE"] = mode
        else: os.environ.pop("PPSFM_BA_SPARSE", None)
        pb = BAProblem(sc)
        st = pb.structure()
        o = bench.opts_fn(10)
        pb.solve(o)
        rates = []
        for r in range(3):
            pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
            t0 = time.perf_counter(); s = pb.solve(o); dt = time.perf_counter() - t0
            rates.append(s.num_iterations / dt)
        print("intrinsics blocks %d, PPSFM_BA_SPARSE=%s: %s -> %.0f LM it/s (solver %d, fallbacks %d)" % (nintr, mode or "1", {k: st[k] for k in ("nnz_used", "tiles", "block_sparse", "chains", "chain_steps")}, max(rates), s.linear_solver, s.cholesky_fallbacks))
        pb.close()
