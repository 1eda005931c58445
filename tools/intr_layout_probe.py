"""Per-image intrinsics beside their pose columns (PrivateIntrinsicsColumns) against the tail layout (PPSFM_BA_INTR_LAYOUT=tail): cfg-3 size, a camera per image with
f and k variable, dense and banded co-visibility - structure, LM it/s, end points of the two layouts.   gpurun -- python tools/intr_layout_probe.py"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem
for window in (40, None):
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, num_intrinsics=500, window=window)
    sc["camera_const_mask"] = np.full(500, 0b0110, dtype=np.uint16)
    ends = {}
    for layout in ("beside", "tail"):
        if layout == "tail": os.environ["PPSFM_BA_INTR_LAYOUT"] = "tail"
        else: os.environ.pop("PPSFM_BA_INTR_LAYOUT", None)
        pb = BAProblem(sc)
        st = pb.structure()
        o = bench.opts_fn(10)
        pb.solve(o)
        rates = []
        for r in range(3):
            pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
            t0 = time.perf_counter(); s = pb.solve(o); dt = time.perf_counter() - t0
            rates.append(s.num_iterations / dt)
        ends[layout] = pb.get_parameters()
        from privacy_preserving_sfm_amd.device import ba_options
        pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
        pb.solve(ba_options(max_num_iterations=10, gradient_tolerance=0.0, phase_timings=1))
        print("   phases [ms per call]:", {k: round(v[0] / max(v[1], 1), 4) for k, v in pb.timings().items()}, flush=True)
        print("window %s, intrinsics %s: %s -> %.0f LM it/s (solver %d, fallbacks %d, final cost %.6g)" % (window, layout, {k: st[k] for k in ("nnz_used", "tiles", "block_sparse", "chains", "chain_steps")}, max(rates), s.linear_solver, s.cholesky_fallbacks, s.final_cost), flush=True)
        pb.close()
    os.environ.pop("PPSFM_BA_INTR_LAYOUT", None)
    print("   layouts agree: poses %.2e points %.2e intrinsics %.2e (relative)" % tuple(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(ends["beside"], ends["tail"])), flush=True)
