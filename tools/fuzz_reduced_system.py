"""Random small scenes (image counts, tracks, windows / clusters, shuffled ids, constant poses / points / tvec components, camera models and masks, shared and
per-image cameras, losses) - the device's reduced camera system and a 4-iteration solve against the CPU oracle, and the wide per-image blocks against the general
lists.   gpurun -- python tools/fuzz_reduced_system.py [cases] [seed]
(round 5, 60 cases, seed 7: every reduced system within 3e-10 of the oracle's; two solves marked CHECK - a camera per image with up to seven free OPENCV
parameters: the two device layouts, assembled and eliminated in different orders, agree on every cost to ten digits while their points differ by 1e-3, and
the oracle leaves them at the third iteration - flat directions of a nearly unobservable problem, not an assembly error.)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options, camera_num_params
orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for case in range(cases):
    C = int(rng.integers(8, 70)); track = int(rng.integers(3, 7)); P = int(rng.integers(8, 40)) * C // 2
    model = int(rng.choice([1, 2, 4])); layout = rng.choice(["fixed", "shared", "per_image"])
    nintr = C if layout == "per_image" else int(rng.integers(1, 4))
    kw = {}
    shape = rng.choice(["dense", "window", "loop", "clusters"])
    if shape == "window" and C >= 24: kw = dict(window=int(rng.integers(max(track + 1, 6), max(track + 2, C // 3))))
    if shape == "loop" and C >= 24: kw = dict(window=int(rng.integers(max(track + 1, 6), max(track + 2, C // 3))), loop=True)
    if shape == "clusters" and C >= 30: kw = dict(clusters=3, bridge=2)
    try:
        sc = synthetic.make_ba_scene(C, P, track, seed=int(rng.integers(1 << 30)), model=model, num_intrinsics=nintr, **kw)
    except Exception as e:
        print("case %d: scene generator refused (%s)" % (case, e)); continue
    if rng.random() < 0.5: sc, _ = synthetic.shuffle_image_ids(sc, seed=int(rng.integers(1 << 30)))
    npar = camera_num_params(model)
    if layout != "fixed":
        while True:
            mask = int(rng.integers(0, 1 << npar))
            if mask != (1 << npar) - 1: break
        sc["camera_const_mask"] = np.full(nintr, mask, dtype=np.uint16)
    for key, frac in (("pose_const", 0.08), ("point_const", 0.05)):
        a = np.ascontiguousarray(sc[key]).copy(); a[rng.random(len(a)) < frac] = 1; sc[key] = a
    tm = np.ascontiguousarray(sc["tvec_const_mask"]).copy(); tm[rng.random(len(tm)) < 0.05] = int(rng.integers(1, 8)); sc["tvec_const_mask"] = tm
    sc["loss_type"] = int(rng.choice([0, 1, 2])); sc["loss_scale"] = 0.05
    radius = float(10.0 ** rng.uniform(0, 4))
    pb = BAProblem(sc)
    S, rhs = pb.reduced_system(radius)
    st = pb.structure()
    s = pb.solve(ba_options(max_num_iterations=4))
    poses, points, intr = pb.get_parameters()
    pb.close()
    ref = orc.ba_reduced_system(sc, radius)
    cols = []
    observed = np.zeros(C, dtype=bool); observed[np.asarray(sc["obs_pose"])] = True
    for c in range(C):
        if sc["pose_const"][c] or not observed[c]: continue      # (an image nothing observes: its columns carry the damping alone on the device, the oracle leaves them out)
        cols += [6 * c, 6 * c + 1, 6 * c + 2] + [6 * c + 3 + j for j in range(3) if not (sc["tvec_const_mask"][c] >> j) & 1]
    ni = S.shape[0] - 6 * C
    # (the device gives every camera an image references its columns - the same on every rank of a group -, the oracle only the cameras that were observed)
    icols, at = [], 6 * C
    if layout != "fixed":
        nv = sum(1 for j in range(npar) if not (mask >> j) & 1)
        seen = np.zeros(nintr, dtype=bool); seen[np.asarray(sc["pose_camera"])[np.asarray(sc["obs_pose"])]] = True
        for k in range(nintr):
            if k in set(int(x) for x in sc["pose_camera"]):
                if seen[k]: icols += list(range(at, at + nv))
                at += nv
        assert at == 6 * C + ni, (at, ni)
    cols = np.array(cols + icols)
    assert len(cols) == ref["nc"], (case, len(cols), ref["nc"])
    eS = np.abs(S[np.ix_(cols, cols)] - ref["S"]).max() / np.abs(ref["S"]).max()
    eb = np.abs(rhs[cols] - ref["rhs"]).max() / max(np.abs(ref["rhs"]).max(), 1e-300)
    rposes, rpoints, rintr, rs, _ = orc.ba_solve(sc, orc.BAOptionsC.defaults(max_num_iterations=4))
    ep = np.abs(points - rpoints).max() / np.abs(rpoints).max(); eq = np.abs(poses - rposes).max() / np.abs(rposes).max()
    same_path = s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    worst = max(worst, eS, eb)
    flag = "" if (eS <= 1e-8 and eb <= 1e-8 and (not same_path or (ep <= 1e-5 and eq <= 1e-5))) else "   <-- CHECK"
    print("case %2d: %2d images %-9s %-8s model %d intr %-9s loss %d  tiles %d/%d chains %d | S %.1e rhs %.1e | solve: poses %.1e points %.1e%s%s" %
          (case, C, shape, "shuffled" if "new_of_old" in sc else "", model, layout, sc["loss_type"], st["nnz_used"], st["tiles"], st["chains"], eS, eb, eq, ep,
           "" if same_path else " (another accept pattern)", flag), flush=True)
    if flag:      # ill-conditioned or wrong?  the same solve with the intrinsics behind the pose columns (round 4's layout, the general lists) and the condition of the system
        os.environ["PPSFM_BA_INTR_LAYOUT"] = "tail"
        pt = BAProblem(sc); st_ = pt.solve(ba_options(max_num_iterations=4)); tposes, tpoints, _ = pt.get_parameters(); ttrace = pt.trace().copy(); pt.close()
        pd_ = BAProblem(sc); pd_.solve(ba_options(max_num_iterations=4)); dtrace = pd_.trace().copy(); pd_.close()
        _, _, _, _, rtrace = orc.ba_solve(sc, orc.BAOptionsC.defaults(max_num_iterations=4))
        print("         costs per iteration  default:", " ".join("%.10e" % v for v in dtrace[:, 0]))
        print("                              tail   :", " ".join("%.10e" % v for v in ttrace[:, 0]))
        print("                              oracle :", " ".join("%.10e" % v for v in rtrace[:, 0]))
        print("         step norms           default:", " ".join("%.6e" % v for v in dtrace[:, 3]), "| oracle:", " ".join("%.6e" % v for v in rtrace[:, 3]))
        os.environ.pop("PPSFM_BA_INTR_LAYOUT")
        w = np.linalg.eigvalsh(ref["S"])
        print("         tail layout against the oracle: poses %.1e points %.1e; against the default layout: poses %.1e points %.1e; condition of the oracle's reduced system %.1e" %
              (np.abs(tposes - rposes).max() / np.abs(rposes).max(), np.abs(tpoints - rpoints).max() / np.abs(rpoints).max(),
               np.abs(tposes - poses).max() / np.abs(poses).max(), np.abs(tpoints - points).max() / np.abs(points).max(), w.max() / max(w.min(), 1e-300)), flush=True)
print("worst relative error of a reduced system: %.2e" % worst)
