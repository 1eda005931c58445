"""Seeded, bounded fuzzing in the GPU suite (round-5 review: the fuzz tools were hand-run; tools/fuzz_reduced_system.py / fuzz_structures.py now share
tests/fuzz_scenes.py with these tests).  Reference behaviour: BundleAdjuster::SetUp / Solve, src/optim/bundle_adjustment.cc:260-542."""
import os

import numpy as np
import pytest

import fuzz_scenes

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _cost_traces_agree(dt, rt, first_tol=1e-7, later_tol=1e-4):
    """cost per iteration while both loops took the same accept / reject decisions: the first trial step's cost to first_tol relative (one linear solve
    from identical inputs), every later one to later_tol of the INITIAL cost (differences of earlier steps are carried and amplified by the problem)"""
    n = min(len(dt), len(rt))
    for i in range(n):
        if dt[i, 6] != rt[i, 6]:
            return True, i
        tol = first_tol * abs(rt[i, 0]) if i <= 1 else later_tol * abs(rt[0, 0])
        if abs(dt[i, 0] - rt[i, 0]) > tol:
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


def _oracle_spread(oracle, sc, iterations, ref_points, ref_poses):
    """how far the ORACLE's own result moves when the initial points are perturbed by 1e-12 relative (three seeds, the largest): the amplification the
    problem itself applies to differences at rounding level - what a device result that solves the same linear systems to 1e-11 is entitled to"""
    spread = 0.0
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        pert = dict(sc, points=np.asarray(sc["points"]) * (1.0 + 1e-12 * rng.uniform(-1, 1, size=np.shape(sc["points"]))))
        pposes, ppoints, _, _, _ = oracle.ba_solve(pert, oracle.BAOptionsC.defaults(max_num_iterations=iterations))
        spread = max(spread, _rel(ppoints, ref_points), _rel(pposes, ref_poses))
    return spread


def _check_case(oracle, sc, m, what):
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    pb = BAProblem(sc)
    S, rhs = pb.reduced_system(m["radius"])
    s = pb.solve(ba_options(max_num_iterations=4))
    poses, points, intr = pb.get_parameters()
    dtrace = pb.trace().copy()
    pb.close()
    ref = oracle.ba_reduced_system(sc, m["radius"])
    cols = fuzz_scenes.oracle_columns(sc, m, S.shape[0])
    assert len(cols) == ref["nc"], what
    assert _rel(S[np.ix_(cols, cols)], ref["S"]) <= 1e-8 and _rel(rhs[cols], ref["rhs"]) <= 1e-8, what      # the assembly: independent of the conditioning
    rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=4))
    ok, upto = _cost_traces_agree(dtrace, rtrace)
    assert ok and upto >= 2, (what, upto, dtrace[:, 0], rtrace[:, 0])
    same_path = s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    drift = max(_rel(points, rpoints), _rel(poses, rposes))
    explained = None
    if same_path and drift > 1e-5:
        # the parameters differ by more than the bar although the systems and the costs agree: acceptable only where the ORACLE ITSELF moves as far when
        # its input changes in the twelfth digit (a badly determined problem: free distortion parameters on a handful of observations per camera)
        explained = _oracle_spread(oracle, sc, 4, rpoints, rposes)
        assert drift <= 20.0 * explained, (what, drift, explained)
    return drift, explained, (poses, points), (rposes, rpoints)


@pytest.mark.parametrize("first", [0, 12])
def test_random_small_scenes_reduced_system_and_solve_match_the_oracle(oracle, first):
    """24 random small scenes (seed 7, cases 0-23: dense / window / loop / cluster co-visibility, shuffled ids, constant poses / points / tvec components,
    three camera models, fixed / shared / per-image intrinsics with random constant masks, three losses): the damped, scaled reduced camera system within
    1e-8 of the oracle's on the oracle's columns; the LM cost trace - the first step to 1e-7, later ones to 1e-4 of the initial cost - while both loops take
    the same decisions; the parameters after four iterations within 1e-5, or within 20 x what the oracle itself moves under a 1e-12 perturbation of its
    input (pinned below for the two cases of this seed where that applies)."""
    from privacy_preserving_sfm_amd.device import camera_num_params
    ran = 0
    for case in range(first, first + 12):
        sc, m = fuzz_scenes.reduced_system_case(7, case, camera_num_params)
        if sc is None:
            continue
        ran += 1
        _check_case(oracle, sc, m, (case, m))
    assert ran >= 9


@pytest.mark.parametrize("case", [36, 50])
def test_badly_determined_problems_move_parameters_not_costs(oracle, case):
    """What tools/fuzz_reduced_system.py flags as CHECK (seed 7, cases 36 and 50: a camera per image with free distortion parameters under a Cauchy loss - round 5's
    tool docstring called such cases "flat directions, not an assembly error" without a test).  Asserted here: (1) the reduced system equals the oracle's to
    1e-8 - the assembly is right whatever the conditioning; (2) the cost traces of the device's two layouts (intrinsics beside their image's pose columns /
    behind all pose columns, PPSFM_BA_INTR_LAYOUT=tail: other kernels, another elimination order) and of the oracle agree - first step 1e-7, later steps
    1e-4 of the initial cost; (3) the parameters differ by MORE than the 1e-5 bar between all three, by comparable amounts (device vs oracle no further
    apart than 20 x the oracle's own movement under a 1e-12 input perturbation): sensitivity of the problem, not a device defect."""
    from privacy_preserving_sfm_amd.device import camera_num_params
    sc, m = fuzz_scenes.reduced_system_case(7, case, camera_num_params)
    assert sc is not None and m["layout"] == "per_image"
    drift, explained, (poses, points), (rposes, rpoints) = _check_case(oracle, sc, m, (case, m))
    assert drift > 1e-5 and explained is not None and explained > 1e-6, (drift, explained)      # (this IS one of the flagged cases; if it stops being one, pin another)
    _, s_t, (tposes, tpoints, _), ttrace = _solve(sc, 4, {"PPSFM_BA_INTR_LAYOUT": "tail"})
    _, s_d, _, dtrace = _solve(sc, 4)
    ok, upto = _cost_traces_agree(ttrace, dtrace)
    assert ok and upto >= 2, (upto, ttrace[:, 0], dtrace[:, 0])
    between_layouts = max(_rel(tpoints, points), _rel(tposes, poses))
    assert between_layouts <= 20.0 * explained and max(_rel(tpoints, rpoints), _rel(tposes, rposes)) <= 20.0 * explained


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
            # (a camera per image with free intrinsics is the badly determined kind of problem pinned above: there the two elimination orders agree on the
            # cost they reach - to 1e-4 of where they started -, not on the digits of the way)
            assert abs(dev[4].final_cost - dense[4].final_cost) <= 1e-4 * dense[4].initial_cost, (case, m)      # (as _cost_traces_agree: to 1e-4 of the initial cost)
            if m["layout"] != "per_image":
                assert max(_rel(a, b) for a, b in zip(dev[3], dense[3])) <= 1e-7, (case, m)
