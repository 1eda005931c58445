"""Seeded, bounded fuzzing in the GPU suite (round-5 review: the fuzz tools were hand-run; tools/fuzz_reduced_system.py / fuzz_structures.py now share
tests/fuzz_scenes.py with these tests).  Reference behaviour: BundleAdjuster::SetUp / Solve, src/optim/bundle_adjustment.cc:260-542."""
import os

import numpy as np
import pytest

import fuzz_scenes

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _cost_traces_agree(dt, rt, tol):
    """cost per iteration while both loops took the same accept / reject decisions"""
    n = min(len(dt), len(rt))
    for i in range(n):
        if dt[i, 6] != rt[i, 6]:
            return True, i
        if abs(dt[i, 0] - rt[i, 0]) > tol * abs(rt[i, 0]):
            return False, i
    return True, n


def _solve(sc, iterations, env=None):
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        pb = BAProblem(sc)
        st = pb.structure()
        s = pb.solve(ba_options(max_num_iterations=iterations))
        out = (st, s, pb.get_parameters(), pb.trace().copy())
        pb.close()
        return out
    finally:
        for k in env:
            os.environ.pop(k)


@pytest.mark.parametrize("first", [0, 12])
def test_random_small_scenes_reduced_system_and_solve_match_the_oracle(oracle, first):
    """24 random small scenes (seed 7, cases 0-23: dense / window / loop / cluster co-visibility, shuffled ids, constant poses / points / tvec components,
    three camera models, fixed / shared / per-image intrinsics with random constant masks, three losses): the damped, scaled reduced camera system within
    1e-8 of the oracle's on the oracle's columns, the LM cost trace within 1e-8 iteration by iteration while both loops take the same decisions, and the
    parameters after four iterations within 1e-5 - or, where a nearly unobservable intrinsics subset lets the PARAMETERS drift along a flat direction, the
    costs still agree (that case is pinned below)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options, camera_num_params
    ran = 0
    for case in range(first, first + 12):
        sc, m = fuzz_scenes.reduced_system_case(7, case, camera_num_params)
        if sc is None:
            continue
        ran += 1
        pb = BAProblem(sc)
        S, rhs = pb.reduced_system(m["radius"])
        s = pb.solve(ba_options(max_num_iterations=4))
        poses, points, intr = pb.get_parameters()
        dtrace = pb.trace().copy()
        pb.close()
        ref = oracle.ba_reduced_system(sc, m["radius"])
        cols = fuzz_scenes.oracle_columns(sc, m, S.shape[0])
        assert len(cols) == ref["nc"], (case, m)
        assert _rel(S[np.ix_(cols, cols)], ref["S"]) <= 1e-8 and _rel(rhs[cols], ref["rhs"]) <= 1e-8, (case, m)
        rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=4))
        ok, upto = _cost_traces_agree(dtrace, rtrace, 1e-8)
        assert ok and upto >= 1, (case, m, upto, dtrace[:, 0], rtrace[:, 0])
        same_path = s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
        if same_path and upto == len(rtrace) and not (_rel(points, rpoints) <= 1e-5 and _rel(poses, rposes) <= 1e-5):
            # the parameters differ although every cost agrees: only acceptable along a flat direction - a reduced system that is numerically singular
            w = np.linalg.eigvalsh(ref["S"])
            assert w.max() / max(w.min(), 1e-300) >= 1e9 and _cost_traces_agree(dtrace, rtrace, 1e-9)[0], (case, m, _rel(points, rpoints), _rel(poses, rposes))
    assert ran >= 9


def test_flat_directions_move_parameters_not_costs(oracle):
    """What tools/fuzz_reduced_system.py's docstring claimed in round 5 ("two CHECK cases: flat directions of a nearly unobservable problem, not an assembly
    error"), as an assertion: a camera per image with every OPENCV parameter but one free (fx fy cx cy k1 k2 p1 p2, a handful of observations per camera: the
    intrinsics and the pose of an image are nearly interchangeable).  The device's two layouts of the reduced system - intrinsics beside their image's pose
    columns (wide blocks) and behind all pose columns (PPSFM_BA_INTR_LAYOUT=tail, the general lists) - and the oracle agree on the COST of every iteration to
    1e-9 (same decisions), and the reduced systems agree to 1e-8; the parameters may differ by more than the 1e-5 bar only because the system's condition
    number exceeds 1e9 - asserted, not assumed."""
    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd.device import BAProblem, camera_num_params
    C = 36
    sc = synthetic.make_ba_scene(C, 400, 4, seed=4242, model=4, num_intrinsics=C, window=9)
    sc["camera_const_mask"] = np.full(C, 0b00000010, dtype=np.uint16)      # only fy constant: seven free parameters per camera
    sc["loss_type"] = 1
    sc["loss_scale"] = 0.05
    st, s, (poses, points, intr), dtrace = _solve(sc, 4)
    _, s_t, (tposes, tpoints, tintr), ttrace = _solve(sc, 4, {"PPSFM_BA_INTR_LAYOUT": "tail"})
    rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=4))
    for a, b, what in ((dtrace, rtrace, "wide vs oracle"), (ttrace, rtrace, "tail vs oracle"), (dtrace, ttrace, "wide vs tail")):
        ok, upto = _cost_traces_agree(a, b, 1e-9)
        assert ok and upto >= 2, (what, upto, a[:, 0], b[:, 0])
    pb = BAProblem(sc)
    S, rhs = pb.reduced_system(1e4)
    pb.close()
    ref = oracle.ba_reduced_system(sc, 1e4)
    m = dict(C=C, layout="per_image", npar=camera_num_params(4), mask=0b10, nintr=C)
    cols = fuzz_scenes.oracle_columns(sc, m, S.shape[0])
    assert _rel(S[np.ix_(cols, cols)], ref["S"]) <= 1e-8 and _rel(rhs[cols], ref["rhs"]) <= 1e-8
    drift = max(_rel(points, rpoints), _rel(poses, rposes), _rel(tpoints, rpoints), _rel(points, tpoints))
    if drift > 1e-5:
        w = np.linalg.eigvalsh(ref["S"])
        assert w.max() / max(w.min(), 1e-300) >= 1e9, (drift, w.max() / w.min())


def test_random_sequence_and_collection_scenes_device_lists_equal_host_lists_and_the_dense_path():
    """Four random mid-size scenes (seed 3: 120-260 images; sequences, loops, clustered collections, shuffled ids; fixed / shared / per-image cameras;
    constant images and points): the block-sparse several-chain path with pair lists and the order's graph built on the device = the same with the host
    builders BIT FOR BIT (structure, reduced system, parameters after three iterations) = the dense path in the caller's order to rounding."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options

    def run(sc, env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            pb = BAProblem(sc)
            st = pb.structure()
            S, rhs = pb.reduced_system(1e3)
            s = pb.solve(ba_options(max_num_iterations=3))
            out = (st, S, rhs, pb.get_parameters(), s)
            pb.close()
            return out
        finally:
            for k in env:
                os.environ.pop(k)

    for case in range(4):
        sc, m = fuzz_scenes.structure_case(3, case, lo=120, hi=260)
        dev = run(sc, {"PPSFM_BA_PAIR_LISTS": "device"})
        host = run(sc, {"PPSFM_BA_PAIR_LISTS": "host"})
        dense = run(sc, {"PPSFM_BA_SPARSE": "0", "PPSFM_BA_ORDERING": "natural"})
        assert dev[0] == host[0] and np.array_equal(dev[1], host[1]) and np.array_equal(dev[2], host[2]), (case, m)
        assert all(np.array_equal(a, b) for a, b in zip(dev[3], host[3])), (case, m)
        assert dev[4].cholesky_fallbacks == 0
        assert _rel(dev[1], dense[1]) <= 1e-9, (case, m)
        if dev[4].num_successful_steps == dense[4].num_successful_steps:
            assert max(_rel(a, b) for a, b in zip(dev[3], dense[3])) <= 1e-7, (case, m)
