"""Seeded random scenes of the fuzz tests (tests/test_gpu_fuzz.py) and tools (tools/fuzz_reduced_system.py, tools/fuzz_structures.py).  Case k of seed s
is a function of (s, k) alone, so a case a tool flags can be pinned in a test by its two numbers."""
import numpy as np

from privacy_preserving_sfm_amd import synthetic


def reduced_system_case(seed, case, camera_num_params):
    """small scenes: 8-70 images, tracks 3-6, dense / window / loop / clusters, shuffled ids, constant poses / points / tvec components, three camera
    models, fixed / shared / per-image intrinsics with a random constant mask, three losses, a random trust-region radius -> (scene, meta) or (None, why)"""
    rng = np.random.default_rng([int(seed), int(case)])
    C = int(rng.integers(8, 70)); track = int(rng.integers(3, 7)); P = int(rng.integers(8, 40)) * C // 2
    model = int(rng.choice([1, 2, 4])); layout = str(rng.choice(["fixed", "shared", "per_image"]))
    nintr = C if layout == "per_image" else int(rng.integers(1, 4))
    kw = {}
    shape = str(rng.choice(["dense", "window", "loop", "clusters"]))
    if shape == "window" and C >= 24: kw = dict(window=int(rng.integers(max(track + 1, 6), max(track + 2, C // 3))))
    if shape == "loop" and C >= 24: kw = dict(window=int(rng.integers(max(track + 1, 6), max(track + 2, C // 3))), loop=True)
    if shape == "clusters" and C >= 30: kw = dict(clusters=3, bridge=2)
    try:
        sc = synthetic.make_ba_scene(C, P, track, seed=int(rng.integers(1 << 30)), model=model, num_intrinsics=nintr, **kw)
    except Exception as e:      # (the generator refuses some combinations: too few images for the clusters asked for, ...)
        return None, str(e)
    shuffled = bool(rng.random() < 0.5)
    if shuffled: sc, _ = synthetic.shuffle_image_ids(sc, seed=int(rng.integers(1 << 30)))
    npar = camera_num_params(model)
    mask = (1 << npar) - 1
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
    return sc, dict(C=C, shape=shape, shuffled=shuffled, model=model, layout=layout, nintr=nintr, npar=npar, mask=mask, radius=radius)


def oracle_columns(sc, meta, n_device):
    """the device's columns of the reduced system that the oracle's system has, in the oracle's order (the device gives every image - and every camera an
    image references - its columns, the same on every rank of a group; the oracle only those that are observed and variable)"""
    C = meta["C"]
    cols = []
    observed = np.zeros(C, dtype=bool); observed[np.asarray(sc["obs_pose"])] = True
    for c in range(C):
        if sc["pose_const"][c] or not observed[c]: continue
        cols += [6 * c, 6 * c + 1, 6 * c + 2] + [6 * c + 3 + j for j in range(3) if not (sc["tvec_const_mask"][c] >> j) & 1]
    ni = n_device - 6 * C
    icols, at = [], 6 * C
    if meta["layout"] != "fixed":
        nv = sum(1 for j in range(meta["npar"]) if not (meta["mask"] >> j) & 1)
        seen = np.zeros(meta["nintr"], dtype=bool); seen[np.asarray(sc["pose_camera"])[np.asarray(sc["obs_pose"])]] = True
        used = set(int(x) for x in sc["pose_camera"])
        for k in range(meta["nintr"]):
            if k in used:
                if seen[k]: icols += list(range(at, at + nv))
                at += nv
        assert at == 6 * C + ni, (at, ni)
    return np.array(cols + icols)


def structure_case(seed, case, lo=120, hi=420):
    """mid-size scenes (sequences, loops, clustered collections, shuffled ids; fixed / shared / per-image cameras; constant images and points)"""
    rng = np.random.default_rng([int(seed), int(case)])
    C = int(rng.integers(lo, hi)); track = int(rng.integers(4, 9)); P = int(rng.integers(20, 60)) * C
    layout = str(rng.choice(["fixed", "shared", "per_image"])); nintr = C if layout == "per_image" else 1
    shape = str(rng.choice(["window", "loop", "clusters"]))
    kw = dict(window=int(rng.integers(12, 40)), loop=bool(shape == "loop")) if shape != "clusters" else dict(clusters=int(rng.integers(3, 6)), bridge=int(rng.integers(2, 5)), topology=str(rng.choice(["star", "chain"])))
    sc = synthetic.make_ba_scene(C, P, track, seed=int(rng.integers(1 << 30)), model=2, num_intrinsics=nintr, **kw)
    if rng.random() < 0.6: sc, _ = synthetic.shuffle_image_ids(sc, seed=int(rng.integers(1 << 30)))
    if layout != "fixed": sc["camera_const_mask"] = np.full(nintr, int(rng.choice([0b0110, 0b0000, 0b1110])), dtype=np.uint16)
    for key, frac in (("pose_const", 0.02), ("point_const", 0.03)):
        a = np.ascontiguousarray(sc[key]).copy(); a[rng.random(len(a)) < frac] = 1; sc[key] = a
    return sc, dict(C=C, shape=shape, layout=layout)
