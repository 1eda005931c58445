"""Random mid-size scenes (120-420 images: sequences, loops, clustered collections, shuffled ids; fixed / shared / per-image cameras; constant images and points):
the block-sparse several-chain path with the lists and the order's graph from the device against (a) the same with the host builders - bit for bit - and
(b) the dense path in the caller's order (PPSFM_BA_SPARSE=0) - to rounding.   gpurun -- python tools/fuzz_structures.py [cases] [seed]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import BAProblem, ba_options
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
def run(sc, env):
    for k, v in env.items(): os.environ[k] = v
    try:
        pb = BAProblem(sc)
        st = pb.structure()
        S, rhs = pb.reduced_system(1e3)
        s = pb.solve(ba_options(max_num_iterations=3))
        out = (st, S, rhs, pb.get_parameters(), s)
        pb.close()
        return out
    finally:
        for k in env: os.environ.pop(k)
bad = 0
for case in range(cases):
    C = int(rng.integers(120, 420)); track = int(rng.integers(4, 9)); P = int(rng.integers(20, 60)) * C
    layout = rng.choice(["fixed", "shared", "per_image"]); nintr = C if layout == "per_image" else 1
    shape = rng.choice(["window", "loop", "clusters"])
    kw = dict(window=int(rng.integers(12, 40)), loop=bool(shape == "loop")) if shape != "clusters" else dict(clusters=int(rng.integers(3, 6)), bridge=int(rng.integers(2, 5)), topology=str(rng.choice(["star", "chain"])))
    sc = synthetic.make_ba_scene(C, P, track, seed=int(rng.integers(1 << 30)), model=2, num_intrinsics=nintr, **kw)
    if rng.random() < 0.6: sc, _ = synthetic.shuffle_image_ids(sc, seed=int(rng.integers(1 << 30)))
    if layout != "fixed": sc["camera_const_mask"] = np.full(nintr, int(rng.choice([0b0110, 0b0000, 0b1110])), dtype=np.uint16)
    for key, frac in (("pose_const", 0.02), ("point_const", 0.03)):
        a = np.ascontiguousarray(sc[key]).copy(); a[rng.random(len(a)) < frac] = 1; sc[key] = a
    dev = run(sc, {"PPSFM_BA_PAIR_LISTS": "device"})
    host = run(sc, {"PPSFM_BA_PAIR_LISTS": "host"})
    dense = run(sc, {"PPSFM_BA_SPARSE": "0", "PPSFM_BA_ORDERING": "natural"})
    same = dev[0] == host[0] and np.array_equal(dev[1], host[1]) and np.array_equal(dev[2], host[2]) and all(np.array_equal(a, b) for a, b in zip(dev[3], host[3]))
    eS = np.abs(dev[1] - dense[1]).max() / np.abs(dense[1]).max()
    ep = max(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) for a, b in zip(dev[3], dense[3]))
    ok = same and eS <= 1e-9 and (ep <= 1e-7 or dev[4].num_successful_steps != dense[4].num_successful_steps) and dev[4].cholesky_fallbacks == 0
    bad += 0 if ok else 1
    print("case %2d: %3d images %6d obs %-8s %-9s | %s tiles %d/%d chains %d steps %d solver %d | device lists = host lists: %s | vs dense path: S %.1e parameters %.1e%s" %
          (case, C, len(sc["obs_pose"]), shape, layout, "renumbered" if dev[0]["reordered"] else "as given  ", dev[0]["nnz_used"], dev[0]["tiles"], dev[0]["chains"], dev[0]["chain_steps"],
           dev[4].linear_solver, same, eS, ep, "" if ok else "   <-- CHECK"), flush=True)
print("%d of %d cases need a look" % (bad, cases))
