"""Random small scenes (tests/fuzz_scenes.py reduced_system_case) - the device's reduced camera system and a 4-iteration solve against the CPU oracle; a case whose
parameters leave the oracle's is looked at again: cost traces of the default layout, the tail layout (PPSFM_BA_INTR_LAYOUT=tail) and the oracle, the condition of
the reduced system.   gpurun -- python tools/fuzz_reduced_system.py [cases] [seed] [first case]
tests/test_gpu_fuzz.py runs a bounded number of these cases in the GPU suite and pins the flagged ones."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
import fuzz_scenes
from privacy_preserving_sfm_amd.device import BAProblem, ba_options, camera_num_params
orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
worst = 0.0
for case in range(first, first + cases):
    sc, m = fuzz_scenes.reduced_system_case(seed, case, camera_num_params)
    if sc is None:
        print("case %d: scene generator refused (%s)" % (case, m)); continue
    pb = BAProblem(sc)
    S, rhs = pb.reduced_system(m["radius"])
    st = pb.structure()
    s = pb.solve(ba_options(max_num_iterations=4))
    poses, points, intr = pb.get_parameters()
    dtrace = pb.trace().copy()
    pb.close()
    ref = orc.ba_reduced_system(sc, m["radius"])
    cols = fuzz_scenes.oracle_columns(sc, m, S.shape[0])
    assert len(cols) == ref["nc"], (case, len(cols), ref["nc"])
    eS = np.abs(S[np.ix_(cols, cols)] - ref["S"]).max() / np.abs(ref["S"]).max()
    eb = np.abs(rhs[cols] - ref["rhs"]).max() / max(np.abs(ref["rhs"]).max(), 1e-300)
    rposes, rpoints, rintr, rs, rtrace = orc.ba_solve(sc, orc.BAOptionsC.defaults(max_num_iterations=4))
    ep = np.abs(points - rpoints).max() / np.abs(rpoints).max(); eq = np.abs(poses - rposes).max() / np.abs(rposes).max()
    same_path = s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    worst = max(worst, eS, eb)
    flag = "" if (eS <= 1e-8 and eb <= 1e-8 and (not same_path or (ep <= 1e-5 and eq <= 1e-5))) else "   <-- CHECK"
    print("case %2d: %2d images %-9s %-8s model %d intr %-9s loss %d  tiles %d/%d chains %d | S %.1e rhs %.1e | solve: poses %.1e points %.1e%s%s" %
          (case, m["C"], m["shape"], "shuffled" if m["shuffled"] else "", m["model"], m["layout"], sc["loss_type"], st["nnz_used"], st["tiles"], st["chains"], eS, eb, eq, ep,
           "" if same_path else " (another accept pattern)", flag), flush=True)
    if flag:      # ill-conditioned or wrong?  the same solve with the intrinsics behind the pose columns (round 4's layout, the general lists) and the condition of the system
        os.environ["PPSFM_BA_INTR_LAYOUT"] = "tail"
        pt = BAProblem(sc); st_ = pt.solve(ba_options(max_num_iterations=4)); tposes, tpoints, _ = pt.get_parameters(); ttrace = pt.trace().copy(); pt.close()
        os.environ.pop("PPSFM_BA_INTR_LAYOUT")
        print("         costs per iteration  default:", " ".join("%.10e" % v for v in dtrace[:, 0]))
        print("                              tail   :", " ".join("%.10e" % v for v in ttrace[:, 0]))
        print("                              oracle :", " ".join("%.10e" % v for v in rtrace[:, 0]))
        print("         step norms           default:", " ".join("%.6e" % v for v in dtrace[:, 3]), "| oracle:", " ".join("%.6e" % v for v in rtrace[:, 3]))
        w = np.linalg.eigvalsh(ref["S"])
        print("         tail layout against the oracle: poses %.1e points %.1e; against the default layout: poses %.1e points %.1e; condition of the oracle's reduced system %.1e" %
              (np.abs(tposes - rposes).max() / np.abs(rposes).max(), np.abs(tpoints - rpoints).max() / np.abs(rpoints).max(),
               np.abs(tposes - poses).max() / np.abs(poses).max(), np.abs(tpoints - points).max() / np.abs(points).max(), w.max() / max(w.min(), 1e-300)), flush=True)
print("worst relative error of a reduced system: %.2e" % worst)
