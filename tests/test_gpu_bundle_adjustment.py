"""K2/K3 + LM driver parity vs the oracle's bundle-adjustment restatement.

L3 contract (BASELINE.json): camera/point parameters within 1e-5 relative of the CPU path after the same
iteration cap.  Because both sides implement the same published LM algorithm in fp64, the per-iteration
costs also agree closely; that is asserted with a looser bound (summation orders differ).
"""
import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu


def _var_cols(sc):
    cols = []
    for c in range(sc["poses"].shape[0]):
        if sc["pose_const"][c]:
            continue
        cols += [6 * c, 6 * c + 1, 6 * c + 2]
        cols += [6 * c + 3 + j for j in range(3) if not (sc["tvec_const_mask"][c] >> j) & 1]
    # variable intrinsics: compact columns after the 6C pose columns, camera by camera (only cameras with observations)
    from privacy_preserving_sfm_amd.device import camera_num_params
    used = set(int(k) for k in np.asarray(sc["pose_camera"]))
    ni = 0
    for k in range(len(sc["camera_model"])):
        if k not in used:
            continue
        ni += sum(1 for j in range(camera_num_params(int(sc["camera_model"][k]))) if not (int(sc["camera_const_mask"][k]) >> j) & 1)
    cols += [6 * sc["poses"].shape[0] + i for i in range(ni)]
    return np.array(cols)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 200, 777])
def test_dense_cholesky_solve(n):
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n + 5))
    A = B @ B.T + 0.5 * np.eye(n)
    # asymmetric right-hand side and a non-symmetric sanity structure in the factor (MFMA layout check)
    b = rng.normal(size=n) * np.arange(1, n + 1)
    x, _ = dense_cholesky_solve(A, b)
    want = np.linalg.solve(A, b)
    assert np.allclose(x, want, rtol=1e-9, atol=1e-9 * np.abs(want).max())


@pytest.mark.parametrize("n", [700, 3001, 5000])
def test_dense_cholesky_task_list_knobs_keep_the_bits(n, monkeypatch):
    """The one-launch factorisation's task list is a schedule, not arithmetic: where whole super-tiles start
    (PPSFM_CHOL_WHOLE_FROM), how far the far updates are deferred (PPSFM_CHOL_SLOPE) and whether a far super-tile takes one or two
    panels per task (PPSFM_CHOL_TWO_PANELS; two: (c - p_k) - p_k+1 in registers instead of a store and a reload) give the same bits."""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, 96))
    A = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n)) * n
    b = rng.normal(size=n)
    monkeypatch.setenv("PPSFM_CHOL_MODE", "tasks")
    x, _ = dense_cholesky_solve(A, b)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
    for env in ({"PPSFM_CHOL_TWO_PANELS": "0"}, {"PPSFM_CHOL_TWO_PANELS": "0", "PPSFM_CHOL_WHOLE_FROM": "12", "PPSFM_CHOL_SLOPE": "0.5"},
                {"PPSFM_CHOL_WHOLE_FROM": "2", "PPSFM_CHOL_SLOPE": "0.2"}, {"PPSFM_CHOL_WHOLE_FROM": "7"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x2, _ = dense_cholesky_solve(A, b)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(x, x2), env


@pytest.mark.parametrize("n", [2944, 3001, 3072, 4500])
def test_dense_cholesky_full_size(n):
    """BASELINE cfg-3's reduced system size and around it: the launch structure changes with the number of block columns
    (deferred pairs of the trailing update start above ~45 block columns, odd / even counts end differently)"""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, 96))
    A = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n)) * n
    b = rng.normal(size=n)
    x, _ = dense_cholesky_solve(A, b)
    r = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    assert r < 1e-12
    x2, _ = dense_cholesky_solve(A, b, repeat=3)       # graph replay gives the same bits
    assert np.array_equal(x, x2)


@pytest.mark.parametrize("n", [200, 255, 700, 2944, 3001, 4500, 6100])
def test_dense_cholesky_task_mode_and_column_mode_agree(n, monkeypatch):
    """PPSFM_CHOL_MODE: "columns" = one launch per block column; "tasks" runs the whole factorisation as ONE launch - a persistent
    chain workgroup plus one workgroup per work item from a priority-sorted list, per-tile dependency counters, mailbox hand-offs
    (unset: tasks up to 128 block columns, columns above).  Same arithmetic per tile in the same order: bitwise equal solutions
    wherever the column mode does not defer trailing updates (up to 48 block columns); beyond that the two orders differ in the
    last bits only.  Replays of the captured graph give the same bits."""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, 96))
    A = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n)) * n
    b = rng.normal(size=n)
    monkeypatch.setenv("PPSFM_CHOL_MODE", "columns")
    x, _ = dense_cholesky_solve(A, b)
    monkeypatch.setenv("PPSFM_CHOL_MODE", "tasks")
    x2, _ = dense_cholesky_solve(A, b)
    x3, _ = dense_cholesky_solve(A, b, repeat=3)
    monkeypatch.delenv("PPSFM_CHOL_MODE")
    x4, _ = dense_cholesky_solve(A, b)                  # the default: one of the two
    assert np.array_equal(x4, x2 if n <= 128 * 64 - 1 else x)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12 and np.linalg.norm(A @ x2 - b) / np.linalg.norm(b) < 1e-12
    assert np.array_equal(x2, x3)
    if n <= 48 * 64 - 1:
        assert np.array_equal(x, x2)
    else:
        assert np.allclose(x, x2, rtol=1e-10, atol=1e-13 * np.abs(x).max())


def test_task_mode_timeout_falls_back_to_column_launches(oracle, monkeypatch):
    """Every wait of the one-launch factorisation is bounded; a timeout (forced here by launching only half of its task list) sets a
    failure bit, and the host repeats the SAME solve / LM step with one launch per block column and stays with that mode."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options, dense_cholesky_solve
    rng = np.random.default_rng(5)
    n = 700
    B = rng.normal(size=(n, 96))
    A = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n)) * n
    b = rng.normal(size=n)
    monkeypatch.setenv("PPSFM_CHOL_MODE", "columns")
    x_ref, _ = dense_cholesky_solve(A, b)
    monkeypatch.setenv("PPSFM_CHOL_MODE", "tasks")
    monkeypatch.setenv("PPSFM_CHOL_TEST_DROP_TASKS", "1")
    x, _ = dense_cholesky_solve(A, b)
    assert np.array_equal(x, x_ref)
    sc = synthetic.make_ba_scene(60, 1500, 6, seed=0xC0FFEE + 9, model=2)      # 361 columns: six block columns
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=4))
    poses, points, _ = pb.get_parameters()
    pb.close()
    assert s.cholesky_fallbacks == 1 and s.linear_solver == 0      # observable: one timeout, per-column launches (PP_LINSOLVE_CHOLESKY_COLUMNS) since
    monkeypatch.delenv("PPSFM_CHOL_TEST_DROP_TASKS")
    monkeypatch.setenv("PPSFM_CHOL_MODE", "columns")
    pb = BAProblem(sc)
    s2 = pb.solve(ba_options(max_num_iterations=4))
    poses2, points2, _ = pb.get_parameters()
    pb.close()
    assert s2.cholesky_fallbacks == 0 and s2.linear_solver == 0
    assert s.num_iterations == s2.num_iterations and s.num_successful_steps == s2.num_successful_steps
    # (the repeated step starts from a re-evaluation at the old point, whose sums are folded in another order than the initial
    # evaluation's: equal to rounding, not bitwise)
    assert np.allclose(poses, poses2, rtol=1e-11, atol=1e-13) and np.allclose(points, points2, rtol=1e-11, atol=1e-13)
    assert abs(s.final_cost - s2.final_cost) <= 1e-8 * abs(s2.final_cost)


def test_dense_cholesky_rejects_indefinite():
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    from privacy_preserving_sfm_amd._capi import PPError
    A = np.eye(100); A[40, 40] = -1.0
    with pytest.raises(PPError):
        dense_cholesky_solve(A, np.ones(100))


@pytest.mark.parametrize("model,loss", [(2, 0), (4, 0), (1, 1), (7, 2)])
def test_reduced_system_matches_oracle(oracle, model, loss):
    from privacy_preserving_sfm_amd.device import BAProblem
    sc = synthetic.make_ba_scene(9, 150, 4, seed=31 + model, model=model)
    sc["loss_type"] = loss
    sc["point_const"][:7] = 1
    pb = BAProblem(sc)
    for radius in (1e4, 3.0):
        S, rhs = pb.reduced_system(radius)
        ref = oracle.ba_reduced_system(sc, radius)
        cols = _var_cols(sc)
        assert len(cols) == ref["nc"]
        Sv = S[np.ix_(cols, cols)]
        scale = np.abs(ref["S"]).max()
        assert np.allclose(Sv, ref["S"], rtol=1e-9, atol=1e-11 * scale)
        assert np.allclose(rhs[cols], ref["rhs"], rtol=1e-9, atol=1e-11 * np.abs(ref["rhs"]).max())
        # constant columns: identity rows, zero rhs
        fixed = np.setdiff1d(np.arange(S.shape[0]), cols)
        assert np.array_equal(S[np.ix_(fixed, fixed)], np.eye(len(fixed))) and np.all(S[np.ix_(fixed, cols)] == 0)
        assert np.all(rhs[fixed] == 0)
    pb.close()


@pytest.mark.parametrize("cams,points,track", [(5, 900, 4), (3, 1400, 3), (6, 300, 5)])
def test_reduced_system_many_observations_per_image(oracle, cams, points, track):
    """Images with 250 / 720 / 1400 observations: the per-image sums of k_reduce take one observation per lane up to 256, two per lane and
    round above it, and loop beyond 512 - every form against the oracle's reduced system, LM diagonal refreshed or kept."""
    from privacy_preserving_sfm_amd.device import BAProblem
    sc = synthetic.make_ba_scene(cams, points, track, seed=77 + cams, model=2)
    per_image = np.bincount(sc["obs_pose"], minlength=cams)
    assert per_image.max() > (512 if points >= 900 else 128)
    pb = BAProblem(sc)
    for radius in (1e4, 0.5):
        S, rhs = pb.reduced_system(radius)
        ref = oracle.ba_reduced_system(sc, radius)
        cols = _var_cols(sc)
        scale = np.abs(ref["S"]).max()
        assert np.allclose(S[np.ix_(cols, cols)], ref["S"], rtol=1e-9, atol=1e-11 * scale)
        assert np.allclose(rhs[cols], ref["rhs"], rtol=1e-9, atol=1e-11 * np.abs(ref["rhs"]).max())
    pb.close()


@pytest.mark.parametrize("model", [2, 1, 4])
def test_cfg1_solve_matches_oracle(oracle, model):
    """BASELINE configs[0]: 20 cams / 2k line obs, single global BA."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(20, 500, 4, seed=0xC0FFEE + 1, model=model)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=1e-10))
    poses, points, intr = pb.get_parameters()
    rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=50, gradient_tolerance=1e-10))
    trace = pb.trace()
    # same LM trajectory: iteration counts and the cost sequence over the first iterations
    k = min(len(trace), len(rtrace), 6)
    assert np.allclose(trace[:k, 0], rtrace[:k, 0], rtol=1e-6, atol=1e-12)
    assert np.array_equal(trace[:k, 6], rtrace[:k, 6])
    assert abs(s.initial_cost - rs.initial_cost) <= 1e-10 * rs.initial_cost
    # L3: parameters within 1e-5 relative
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max()
    assert np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    # and both sit on the ground truth (noise-free data, gauge fixed)
    assert np.abs(points - sc["gt_points"]).max() < 1e-6
    assert np.abs(poses[:, 4:] - sc["gt_poses"][:, 4:]).max() < 1e-6
    assert np.array_equal(poses[0], sc["poses"][0]) and poses[1, 4] == sc["poses"][1, 4]
    assert np.array_equal(intr, sc["intr"])
    assert s.num_residuals == 4000 and s.num_effective_parameters == 19 * 6 - 1 + 1500
    pb.close()


@pytest.mark.parametrize("seed,scale", [(5, 30.0), (6, 60.0), (7, 100.0)])
def test_rejected_steps_follow_the_oracle(oracle, seed, scale):
    """a start far from the optimum: the trust region has to shrink, so the run contains REJECTED trial steps.  The device
    loop enqueues the accept path speculatively and must undo it (old point current again, its Jacobians restored): the
    sequence of successful / unsuccessful steps, costs and the final parameters are the oracle's."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(8, 160, 4, seed=seed, model=2)
    rng = np.random.default_rng(seed)
    sc["points"] = sc["gt_points"] + scale * 1e-2 * rng.normal(size=sc["gt_points"].shape)
    sc["poses"][2:, 4:] += scale * 2e-3 * rng.normal(size=sc["poses"][2:, 4:].shape)
    pb = BAProblem(sc)
    opts = dict(max_num_iterations=40, gradient_tolerance=1e-10)
    s = pb.solve(ba_options(**opts))
    poses, points, _ = pb.get_parameters()
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    trace = pb.trace()
    assert rs.num_unsuccessful_steps > 0, "scene does not exercise rejection"
    # identical accept / reject pattern up to the point where costs reach rounding level
    k = min(len(trace), len(rtrace))
    big = rtrace[:k, 0] > 1e-12 * rtrace[0, 0]
    assert np.array_equal(trace[:k, 6][big], rtrace[:k, 6][big])
    assert np.allclose(trace[:k, 0][big], rtrace[:k, 0][big], rtol=1e-6)
    # trust-region radius: radius / max(1/3, 1 - (2 rho - 1)^3) amplifies the rounding of rho = cost change / model change wherever the
    # clamp is not active (observed: 1.2e-9 between two summation orders of the same sums)
    assert np.allclose(trace[:k, 5][big], rtrace[:k, 5][big], rtol=1e-7)
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max()
    assert np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    # the state after the solve is a consistent evaluation point: re-evaluating gives the reported final cost
    cost = pb.evaluate()[0]
    assert abs(cost - s.final_cost) <= 1e-9 * max(s.final_cost, 1e-300) + 1e-18
    pb.close()


@pytest.mark.parametrize("loss", [0, 1, 2])
def test_fused_trial_cost_is_the_separate_launch_bit_for_bit(loss, monkeypatch):
    """The cost at the trial point is evaluated inside k_model_cost_apply (every observation applies the step to its own pose and point)
    instead of by a k_line_eval<0> launch over the stored trial point (PPSFM_BA_FUSED_TRIAL_COST=0): the same arithmetic and block sums,
    so the whole trace - costs, radii, accept / reject pattern, rejected steps included - and the parameters are bitwise equal."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    monkeypatch.setenv("PPSFM_BA_FUSED_STEP", "0")      # (the per-observation kernels; k_step_points sums in another association)
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("PPSFM_BA_FUSED_TRIAL_COST", fused)
        sc = synthetic.make_ba_scene(12, 300, 5, seed=21, model=2)
        rng = np.random.default_rng(21)
        sc["points"] = sc["gt_points"] + 0.4 * rng.normal(size=sc["gt_points"].shape)
        sc["loss_type"] = loss; sc["loss_scale"] = 0.7
        pb = BAProblem(sc)
        s = pb.solve(ba_options(max_num_iterations=25))
        poses, points, _ = pb.get_parameters()
        runs.append((pb.trace().copy(), poses.copy(), points.copy(), s.num_unsuccessful_steps))
        pb.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    assert len(runs[0][0]) > 3


@pytest.mark.parametrize("track", [4, 8, 11])
def test_fused_point_step_follows_the_two_kernel_form(track, monkeypatch):
    """k_step_points (point steps, model cost change, trial point and its cost in one pass over the observations, four lanes per point)
    against k_backsub_points + k_model_cost_apply (PPSFM_BA_FUSED_STEP=0): the same steps and trial points, the two sums in a different
    (equally fixed) association - traces agree to rounding, accept / reject pattern and parameters alike.  11 observations per point:
    a lane's third observation takes the reload path."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("PPSFM_BA_FUSED_STEP", fused)
        sc = synthetic.make_ba_scene(14, 260, track, seed=33 + track, model=2)
        rng = np.random.default_rng(33)
        sc["points"] = sc["gt_points"] + 0.3 * rng.normal(size=sc["gt_points"].shape)
        sc["point_const"][:5] = 1
        sc["loss_type"] = 1; sc["loss_scale"] = 0.8
        pb = BAProblem(sc)
        s = pb.solve(ba_options(max_num_iterations=20))
        poses, points, _ = pb.get_parameters()
        runs.append((pb.trace().copy(), poses.copy(), points.copy(), s))
        pb.close()
    (tf, pf, xf, sf), (t0, p0, x0, s0) = runs
    # (once the cost sits at its rounding floor a trial step can change it by exactly zero, which ends a solve without tolerances as "converged":
    # where that happens is each form's rounding - the traces are compared while the cost still moves)
    n = min(len(tf), len(t0))
    assert n > 3
    tf, t0 = tf[:n], t0[:n]
    big = t0[:, 0] > 1e-12 * t0[0, 0]
    # (a robust loss leaves a non-zero minimum: once the cost moves by less than 1e-9 of itself the gain ratio - and with it the next radius - is
    # rounding as well)
    moving = np.concatenate([[True], np.abs(np.diff(t0[:, 0])) > 1e-9 * t0[1:, 0]])
    big &= np.cumprod(moving).astype(bool)
    assert big.sum() > 3
    assert np.array_equal(tf[big, 6], t0[big, 6]) and np.allclose(tf[big, 0], t0[big, 0], rtol=1e-10) and np.allclose(tf[big, 5], t0[big, 5], rtol=1e-7)
    assert np.abs(pf - p0).max() <= 1e-7 * np.abs(p0).max() and np.abs(xf - x0).max() <= 1e-7 * np.abs(x0).max()
    assert np.array_equal(xf[:5], sc["points"][:5])      # constant points did not move


def test_tolerance_terminations_leave_the_accepted_point(oracle):
    """function / parameter tolerance fire on a trial step that is NOT applied (Ceres checks them before accepting): the
    parameters after the solve are those of the last accepted step, also when the accept path had been enqueued already."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(10, 200, 4, seed=21, model=2)
    rng = np.random.default_rng(3)
    sc["lines"][:, 2] += 2e-4 * rng.normal(size=len(sc["lines"]))      # measurement noise: the optimum has a non-zero cost, so the
    for opts in (dict(max_num_iterations=50, function_tolerance=1e-6), dict(max_num_iterations=50, parameter_tolerance=1e-6),   # tolerances act above rounding level
                 dict(max_num_iterations=50, gradient_tolerance=1e-3), dict(max_num_iterations=3)):
        pb = BAProblem(sc)
        s = pb.solve(ba_options(**opts))
        poses, points, _ = pb.get_parameters()
        rposes, rpoints, _, rs, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
        assert (s.num_iterations, s.num_successful_steps, s.termination) == (rs.num_iterations, rs.num_successful_steps, rs.termination), opts
        assert np.abs(points - rpoints).max() <= 1e-7 * np.abs(rpoints).max(), opts
        assert np.abs(poses - rposes).max() <= 1e-7 * np.abs(rposes).max(), opts
        cost = pb.evaluate()[0]
        assert abs(cost - s.final_cost) <= 1e-9 * max(s.final_cost, 1e-300) + 1e-18, opts
        pb.close()


def test_local_ba_preset_soft_l1_with_constant_blocks(oracle):
    """local-BA shaped problem: SOFT_L1 loss, constant pose + constant tvec.x, some constant points
    (sfm/incremental_mapper.cc:828-854), gradient tolerance 10, 25 iterations."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(6, 200, 3, seed=77, model=2)
    sc["loss_type"] = 1
    sc["point_const"][::9] = 1
    sc["points"][::9] = sc["gt_points"][::9]
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=25, gradient_tolerance=10.0))
    poses, points, _ = pb.get_parameters()
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=25, gradient_tolerance=10.0))
    assert s.num_iterations == rs.num_iterations and s.termination == rs.termination
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max()
    assert np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    assert np.array_equal(points[::9], sc["points"][::9])
    assert abs(s.final_cost - rs.final_cost) <= 1e-6 * max(rs.final_cost, 1e-12) + 1e-12
    pb.close()


def test_cfg2_size_solve_properties():
    """BASELINE configs[1] size (100 cams / 40k obs): converges to the ground truth, deterministic."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(100, 5000, 8, seed=0xC0FFEE + 2, model=2)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=1e-8))
    poses, points, _ = pb.get_parameters()
    assert s.final_cost < 1e-12 * s.initial_cost + 1e-14
    assert np.abs(points - sc["gt_points"]).max() < 1e-6
    assert np.abs(poses[:, 4:] - sc["gt_poses"][:, 4:]).max() < 1e-6
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    s2 = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=1e-8))
    poses2, points2, _ = pb.get_parameters()
    assert np.array_equal(poses, poses2) and np.array_equal(points, points2) and s2.num_iterations == s.num_iterations
    pb.close()


def test_cfg3_full_size_solve_properties():
    """BASELINE configs[2] at full size (500 cams / 200k obs, the bench workload): size-independent properties — the
    noise-free problem converges to the ground truth under the reference's gauge, every LM step of the run is successful,
    a second run from the same start reproduces the parameters bit for bit, and the reported cost is the cost at the
    returned parameters."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=1e-8))
    poses, points, _ = pb.get_parameters()
    assert s.num_residuals == 400000 and s.num_effective_parameters == 499 * 6 - 1 + 75000
    assert s.num_unsuccessful_steps == 0 and s.termination == 0
    assert s.final_cost < 1e-12 * s.initial_cost
    assert np.abs(points - sc["gt_points"]).max() < 1e-6
    assert np.abs(poses[:, 4:] - sc["gt_poses"][:, 4:]).max() < 1e-6
    assert np.array_equal(poses[0], sc["poses"][0]) and poses[1, 4] == sc["poses"][1, 4]       # the gauge blocks did not move
    assert abs(pb.evaluate()[0] - s.final_cost) <= 1e-9 * s.final_cost + 1e-18
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    s2 = pb.solve(ba_options(max_num_iterations=50, gradient_tolerance=1e-8))
    poses2, points2, _ = pb.get_parameters()
    assert np.array_equal(poses, poses2) and np.array_equal(points, points2) and s2.num_iterations == s.num_iterations
    pb.close()


def _intr_scene(num_cams, num_points, track, model, num_intrinsics, const_bits, seed):
    """scene with variable intrinsics: `const_bits` = parameters held constant (SubsetParameterization), start
    intrinsics perturbed by ~1 %"""
    sc = synthetic.make_ba_scene(num_cams, num_points, track, seed=seed, model=model, num_intrinsics=num_intrinsics)
    sc["camera_const_mask"] = np.full(num_intrinsics, const_bits, dtype=np.uint16)
    rng = np.random.default_rng(seed + 1)
    from privacy_preserving_sfm_amd.device import camera_num_params
    npar = camera_num_params(model)
    intr = np.array(sc["intr"], dtype=np.float64).copy()
    for k in range(num_intrinsics):
        for j in range(npar):
            if not (const_bits >> j) & 1:
                intr[k, j] *= 1.0 + 0.01 * rng.normal() if abs(intr[k, j]) > 1e-6 else 1.0
                if abs(intr[k, j]) <= 1e-6:
                    intr[k, j] = 1e-3 * rng.normal()
    sc["intr"] = intr
    return sc


# (model, intrinsics blocks, constant-parameter bits): SIMPLE_RADIAL f+k shared / per image; PINHOLE focal only;
# OPENCV everything but the principal point; one block with every parameter variable
@pytest.mark.parametrize("model,nintr,const_bits", [(2, 1, 0b0110), (2, 9, 0b0110), (1, 3, 0b1100), (4, 2, 0b00001100), (2, 1, 0)])
def test_reduced_system_with_variable_intrinsics_matches_oracle(oracle, model, nintr, const_bits):   # bundle_adjustment.cc:490-528
    from privacy_preserving_sfm_amd.device import BAProblem
    sc = _intr_scene(9, 160, 4, model, nintr, const_bits, seed=77 + model + nintr)
    sc["point_const"][:9] = 1
    pb = BAProblem(sc)
    for radius in (1e4, 2.0):
        S, rhs = pb.reduced_system(radius)
        ref = oracle.ba_reduced_system(sc, radius)
        cols = _var_cols(sc)
        assert len(cols) == ref["nc"] and S.shape[0] == 6 * 9 + (len(cols) - len([c for c in cols if c < 54]))
        Sv = S[np.ix_(cols, cols)]
        scale = np.abs(ref["S"]).max()
        assert np.allclose(Sv, ref["S"], rtol=1e-9, atol=1e-11 * scale)
        assert np.allclose(rhs[cols], ref["rhs"], rtol=1e-9, atol=1e-11 * np.abs(ref["rhs"]).max())
    pb.close()


@pytest.mark.parametrize("model,nintr,const_bits", [(2, 1, 0b0110), (2, 20, 0b0110), (4, 2, 0b00001100)])
def test_solve_with_variable_intrinsics_matches_oracle(oracle, model, nintr, const_bits):
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = _intr_scene(20, 500, 5, model, nintr, const_bits, seed=0xC0FFEE + 11 * model + nintr)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=40, gradient_tolerance=1e-10))
    poses, points, intr = pb.get_parameters()
    rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=40, gradient_tolerance=1e-10))
    trace = pb.trace()
    k = min(len(trace), len(rtrace), 6)
    assert np.allclose(trace[:k, 0], rtrace[:k, 0], rtol=1e-6, atol=1e-12)
    assert np.array_equal(trace[:k, 6], rtrace[:k, 6])
    assert s.final_cost < 1e-3 * s.initial_cost
    assert np.abs(intr - rintr).max() <= 1e-5 * np.abs(rintr).max()
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max()
    assert np.abs(poses - rposes).max() <= 1e-5
    # constant parameters did not move
    start = np.asarray(sc["intr"])
    for j in range(12):
        if (const_bits >> j) & 1:
            assert np.array_equal(intr[:, j], start[:, j])
    pb.close()


@pytest.mark.parametrize("nintr", [1, 300])
def test_sequence_scene_with_variable_intrinsics_takes_the_block_sparse_path(oracle, monkeypatch, nintr):
    """refine_focal_length / refine_extra_params (bundle_adjustment.cc:490-528) on a sequence scene: the intrinsics rows of the reduced camera system are
    dense, the pose part keeps its band - an arrow.  The images are dissected as without them (the intrinsics columns stay behind the pose columns: part
    of the last separator), the factorisation runs several chains; same trajectory as the dense path (PPSFM_BA_SPARSE=0, the caller's order) to rounding
    and as the oracle (BASELINE's tolerance).  One camera shared by all images, and a camera per image (300 x 2 more columns) - whose variable intrinsics sit
    BESIDE their image's pose columns in the reduced system (round 5, PrivateIntrinsicsColumns: they couple with the same images as the pose and belong to its part
    of the dissection; behind all pose columns - PPSFM_BA_INTR_LAYOUT=tail, round 4's layout - they are dense block rows every chain waits for): fewer chain steps,
    the same solve."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(300, 8000, 6, seed=0xC0FFEE + 57, model=2, window=20, num_intrinsics=nintr)
    sc["camera_const_mask"] = np.full(nintr, 0b0110, dtype=np.uint16)      # f and k of SIMPLE_RADIAL variable
    opts = dict(max_num_iterations=6)
    pb = BAProblem(sc)
    st = pb.structure()
    assert st["block_sparse"] and st["reordered"] and st["chains"] >= 2, st
    s = pb.solve(ba_options(**opts))
    poses, points, intr = pb.get_parameters()
    S, rhs = pb.reduced_system(1e4)
    pb.close()
    assert s.linear_solver == 2 and s.cholesky_fallbacks == 0
    monkeypatch.setenv("PPSFM_BA_SPARSE", "0")
    pd = BAProblem(sc)
    assert not pd.structure()["block_sparse"] and not pd.structure()["reordered"]
    sd = pd.solve(ba_options(**opts))
    dposes, dpoints, dintr = pd.get_parameters()
    Sd, rhsd = pd.reduced_system(1e4)
    pd.close()
    monkeypatch.delenv("PPSFM_BA_SPARSE")
    # (the sums of a camera's intrinsics rows run over its observations in the handle's image order: another association, equal to rounding)
    assert np.abs(S - Sd).max() <= 1e-9 * np.abs(Sd).max() and np.abs(rhs - rhsd).max() <= 1e-9 * np.abs(rhsd).max()
    assert s.num_iterations == sd.num_iterations and s.num_successful_steps == sd.num_successful_steps
    assert np.abs(poses - dposes).max() <= 1e-8 * np.abs(dposes).max() and np.abs(points - dpoints).max() <= 1e-8 * np.abs(dpoints).max()
    assert np.abs(intr - dintr).max() <= 1e-8 * np.abs(dintr).max()
    rposes, rpoints, rintr, rs, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    assert np.abs(intr - rintr).max() <= 1e-5 * np.abs(rintr).max()
    if nintr > 1:
        monkeypatch.setenv("PPSFM_BA_INTR_LAYOUT", "tail")
        pt = BAProblem(sc)
        stt = pt.structure()
        s_t = pt.solve(ba_options(**opts))
        tposes, tpoints, tintr = pt.get_parameters()
        St, rhst = pt.reduced_system(1e4)
        pt.close()
        monkeypatch.delenv("PPSFM_BA_INTR_LAYOUT")
        assert stt["block_sparse"] and st["chain_steps"] < stt["chain_steps"] and st["nnz_used"] < stt["nnz_used"], (st, stt)
        assert np.abs(S - St).max() <= 1e-9 * np.abs(St).max() and np.abs(rhs - rhst).max() <= 1e-9 * np.abs(rhst).max()      # the same system, column by column
        assert s.num_iterations == s_t.num_iterations and s.num_successful_steps == s_t.num_successful_steps
        assert np.abs(poses - tposes).max() <= 1e-8 * np.abs(tposes).max() and np.abs(points - tpoints).max() <= 1e-8 * np.abs(tpoints).max()
        assert np.abs(intr - tintr).max() <= 1e-8 * np.abs(tintr).max()


@pytest.mark.parametrize("model,const_bits,nv", [(2, 0b0110, 2), (2, 0, 4), (1, 0b1100, 2), (4, 0b00001100, 6), (4, 0, 8)])
def test_camera_per_image_wide_blocks_equal_the_general_block_pairs(oracle, monkeypatch, model, const_bits, nv):
    """A camera per image with n_v variable parameters beside its pose columns: the image's 6 + n_v columns are assembled as ONE block by the pose gather with
    wider rows (ba_solver.hip k_schur_wide_self / k_schur_wide_pairs, n_v = 2, 4, 6, 8) instead of the block-pair lists built for shared cameras
    (ba_intr.hip; PPSFM_BA_INTR_WIDE=0 keeps them).  The same reduced system to rounding - with constant poses in the middle of the sequence (their
    intrinsics still couple with their neighbours': every image is listed), constant points (direct terms only) and a point seen twice by one image -
    and the oracle's system and solve (bundle_adjustment.cc:490-528)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    C = 40
    sc = _intr_scene(C, 1200, 5, model, C, const_bits, seed=0xC0FFEE + 101 * model + nv)
    sc["pose_const"] = np.ascontiguousarray(sc["pose_const"]).copy(); sc["pose_const"][[7, 23]] = 1
    sc["point_const"] = np.ascontiguousarray(sc["point_const"]).copy(); sc["point_const"][::37] = 1
    sc["obs_pose"] = np.ascontiguousarray(sc["obs_pose"]).copy()
    sc["obs_pose"][5] = sc["obs_pose"][6]      # point 1 is seen twice by one image
    out = {}
    for wide in ("1", "0"):
        monkeypatch.setenv("PPSFM_BA_INTR_WIDE", wide)
        pb = BAProblem(sc)
        S, rhs = pb.reduced_system(1e3)
        s = pb.solve(ba_options(max_num_iterations=5))
        out[wide] = (S, rhs, pb.get_parameters(), s)
        pb.close()
    monkeypatch.delenv("PPSFM_BA_INTR_WIDE")
    (S, rhs, (poses, points, intr), s), (S0, rhs0, (poses0, points0, intr0), s0) = out["1"], out["0"]
    assert S.shape[0] == 6 * C + nv * C
    assert np.abs(S - S0).max() <= 1e-10 * np.abs(S0).max() and np.abs(rhs - rhs0).max() <= 1e-10 * np.abs(rhs0).max()
    assert s.num_iterations == s0.num_iterations and s.num_successful_steps == s0.num_successful_steps
    assert np.abs(poses - poses0).max() <= 1e-8 * np.abs(poses0).max() and np.abs(points - points0).max() <= 1e-8 * np.abs(points0).max()
    assert np.abs(intr - intr0).max() <= 1e-8 * np.abs(intr0).max()
    ref = oracle.ba_reduced_system(sc, 1e3)
    cols = _var_cols(sc)
    assert len(cols) == ref["nc"], (len(cols), ref["nc"])
    scale = np.abs(ref["S"]).max()
    assert np.allclose(S[np.ix_(cols, cols)], ref["S"], rtol=1e-8, atol=1e-10 * scale)
    assert np.allclose(rhs[cols], ref["rhs"], rtol=1e-8, atol=1e-10 * np.abs(ref["rhs"]).max())


def _filter_scene(seed, n_intr=1):
    """BA scene for the filters: long tracks, ~half of the lines gravity-aligned, some observations corrupted, a few
    points behind a camera, a few points with a tiny baseline (far away), image bounds that cut some projections"""
    sc = synthetic.make_ba_scene(14, 600, 6, seed=seed, model=2, num_intrinsics=n_intr, noise_point=0.0, noise_q=0.0, noise_t=0.0)
    rng = np.random.default_rng(seed)
    M = len(sc["obs_pose"])
    lines = sc["lines"].copy()
    bad = rng.choice(M, M // 12, replace=False)
    lines[bad, 2] += rng.normal(0, 0.02, len(bad))                 # corrupted line offsets -> large pixel error
    sc["lines"] = lines
    pts = sc["points"].copy()
    pts[:12] *= 40.0                                               # far points: small triangulation angles
    pts[12:20] = -pts[12:20] - np.array([0, 0, 12.0])              # behind the cameras
    sc["points"] = pts
    aligned = rng.random(M) < 0.5
    aligned[np.isin(sc["obs_point"], np.arange(20, 30))] = True    # tracks with aligned lines only
    f = float(sc["intr"][0, 0])
    cam_size = np.tile(np.array([[int(2.2 * f), int(1.8 * f)]], dtype=np.int32), (n_intr, 1))
    return sc, aligned, cam_size


@pytest.mark.parametrize("seed,max_err,min_ang,subset", [(1, 4.0, 1.5, False), (2, 1.0, 0.5, True), (3, 12.0, 6.0, False)])
def test_filter_points3d_matches_oracle(oracle, seed, max_err, min_ang, subset):      # base/reconstruction.cc:425-439, 594-719
    from privacy_preserving_sfm_amd.device import BAProblem
    sc, aligned, cam_size = _filter_scene(seed, n_intr=2 if seed == 3 else 1)
    sub = (np.arange(600) % 3 != 0) if subset else None
    pb = BAProblem(sc)
    rep, od, pd, pe = pb.filter_points(max_err, min_ang, cam_size, obs_aligned=aligned, point_subset=sub)
    rnf, rod, rpd, rpe = oracle.filter_points3d(sc, max_err, min_ang, cam_size, aligned, sub)
    assert rep.num_filtered == rnf and np.array_equal(od, rod) and np.array_equal(pd, rpd)
    assert np.allclose(pe, rpe, rtol=1e-9, atol=1e-12)
    assert rep.num_points_deleted == int(rpd.sum()) and rep.num_observations_deleted == int(rod.sum())
    # every rule fires somewhere in this scene
    assert rpd.sum() > 20 and (~rpd).sum() > 100 and (rod & ~rpd[sc["obs_point"]]).sum() > 10
    n, neg = pb.filter_negative_depth()
    rn, rneg = oracle.filter_negative_depth(sc)
    assert n == rn and np.array_equal(neg, rneg) and n >= 8 * 6
    pb.close()


def test_filter_points3d_full_size_properties():
    """BASELINE configs[2] shape (500 cams / 200k observations): idempotence — a second pass over the surviving tracks
    with the same thresholds removes nothing more except through the shorter-track rules — and consistency of the masks"""
    from privacy_preserving_sfm_amd.device import BAProblem
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)
    rng = np.random.default_rng(0)
    M = len(sc["obs_pose"])
    lines = sc["lines"].copy()
    bad = rng.choice(M, M // 50, replace=False)
    lines[bad, 2] += rng.normal(0, 0.05, len(bad))
    sc["lines"] = lines
    f = float(sc["intr"][0, 0])
    cam_size = np.array([[int(4 * f), int(4 * f)]], dtype=np.int32)
    pb = BAProblem(sc)
    rep, od, pd, pe = pb.filter_points(4.0, 1.5, cam_size)
    assert rep.num_observations_deleted == od.sum() and rep.num_points_deleted == pd.sum()
    assert od[pd[sc["obs_point"]]].all()                            # a deleted point takes its whole track
    kept = ~pd
    assert (pe[kept] >= 0).all() and (pe[kept] <= 4.0).all()
    surv = np.bincount(sc["obs_point"][~od], minlength=25000)
    assert (surv[kept] >= 4).all()                                  # >= 4 observations survive on every kept point (:705)
    assert 0 < od.sum() < M // 4
    pb.close()


def test_iteration_callback_sees_every_iteration_and_can_stop(oracle):
    """ceres::IterationCallback (the reference registers BundleAdjustmentIterationCallback, controllers/bundle_adjustment.cc:43-61,
    87-88: SOLVER_TERMINATE_SUCCESSFULLY once its thread is stopped): called after every iteration incl. iteration 0 with the
    iteration summary; SOLVER_ABORT -> USER_FAILURE, SOLVER_TERMINATE_SUCCESSFULLY -> USER_SUCCESS; the LM trajectory is unchanged by observing it."""
    from privacy_preserving_sfm_amd import _capi
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(12, 300, 4, seed=91, model=2)
    pb = BAProblem(sc)
    s0 = pb.solve(ba_options(max_num_iterations=6))
    t0 = pb.trace()
    p0 = pb.get_parameters()
    seen = []
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    s1 = pb.solve(ba_options(max_num_iterations=6), iteration_callback=lambda it: seen.append(
        (it.iteration, it.step_is_successful, it.cost, it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius)) or 0)
    t1 = pb.trace()
    p1 = pb.get_parameters()
    assert s1.num_iterations == s0.num_iterations == 6 and np.array_equal(t0, t1)
    assert all(np.array_equal(a, b) for a, b in zip(p0, p1))
    assert [r[0] for r in seen] == list(range(7))
    got = np.array([[r[2], r[3], r[4], r[5], r[6], r[7], r[1]] for r in seen])
    assert np.array_equal(got, t1)                                   # cost and gradient norm AT the accepted point, as Ceres reports them
    # abort after the second iteration: USER_FAILURE, two iterations done, device state = the last accepted point
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    s2 = pb.solve(ba_options(max_num_iterations=6), iteration_callback=lambda it: _capi.SOLVER_ABORT if it.iteration == 2 else _capi.SOLVER_CONTINUE)
    assert s2.termination == _capi.TERM_USER_FAILURE and s2.num_iterations == 2
    assert s2.final_cost == t1[2, 0]
    assert abs(pb.evaluate()[0] - s2.final_cost) <= 1e-9 * s2.final_cost + 1e-18
    pb.set_parameters(sc["poses"], sc["points"], sc["intr"])
    s3 = pb.solve(ba_options(max_num_iterations=6), iteration_callback=lambda it: _capi.SOLVER_TERMINATE_SUCCESSFULLY)
    assert s3.termination == _capi.TERM_USER_SUCCESS and s3.num_iterations == 0 and s3.final_cost == s3.initial_cost
    pb.close()
    # through the BundleAdjuster mirror: an abort leaves the reconstruction as it was given (Summary::IsSolutionUsable() false)
    from privacy_preserving_sfm_amd.bundle_adjustment import BundleAdjuster, BundleAdjustmentConfig, BundleAdjustmentOptions, Reconstruction
    rec = Reconstruction.from_scene(sc)
    cfg = BundleAdjustmentConfig()
    for i in range(12):
        cfg.AddImage(i)
    cfg.SetConstantPose(0)
    cfg.SetConstantTvec(1, [0])
    opt = BundleAdjustmentOptions()
    opt.print_summary = False
    opt.solver_options.max_num_iterations = 6
    opt.solver_options.iteration_callback = lambda it: _capi.SOLVER_ABORT if it.iteration == 1 else 0
    before = np.array([rec.Point3D(p).xyz for p in range(300)])
    ba = BundleAdjuster(opt, cfg)
    assert ba.Solve(rec) is True and ba.Summary().termination == _capi.TERM_USER_FAILURE
    assert np.array_equal(before, np.array([rec.Point3D(p).xyz for p in range(300)]))


def test_numeric_failure_returns_a_summary():
    """a problem whose every step is invalid (NaN line): pp_ba_solve reports FAILURE with a filled summary, the Python
    mirror keeps it as Summary() and leaves the parameters alone"""
    from privacy_preserving_sfm_amd import _capi
    from privacy_preserving_sfm_amd.bundle_adjustment import BundleAdjuster, BundleAdjustmentConfig, BundleAdjustmentOptions, Reconstruction
    sc = synthetic.make_ba_scene(6, 60, 3, seed=5, model=2)
    sc["points"][7] = sc["poses"][2, 4:] * 0 + np.array([np.nan, 0.0, 1.0])
    rec = Reconstruction.from_scene(sc)
    cfg = BundleAdjustmentConfig()
    for i in range(6):
        cfg.AddImage(i)
    cfg.SetConstantPose(0)
    opt = BundleAdjustmentOptions()
    opt.print_summary = False
    ba = BundleAdjuster(opt, cfg)
    assert ba.Solve(rec) is True
    s = ba.Summary()
    assert s is not None and s.termination == _capi.TERM_FAILURE and s.num_residuals == 2 * len(sc["obs_pose"])


def test_create_refuses_pair_lists_beyond_32_bits():
    """one point seen by 47 000 variable images: ~2.2e9 Schur pair entries, which the 32-bit list offsets cannot hold
     - pp_ba_create must say so instead of building corrupt lists"""
    from privacy_preserving_sfm_amd._capi import PPError, PP_ERR_INVALID
    from privacy_preserving_sfm_amd.device import BAProblem
    C = 47000
    sc = dict(lines=np.tile([1.0, 0.0, 0.0], (C, 1)), obs_pose=np.arange(C, dtype=np.int32), obs_point=np.zeros(C, dtype=np.int32),
              pose_camera=np.zeros(C, dtype=np.int32), camera_model=np.array([2], dtype=np.int32), poses=np.tile([1.0, 0, 0, 0, 0, 0, 0], (C, 1)),
              points=np.array([[0.0, 0.0, 5.0]]), intr=np.array([[1000.0, 640, 480, 0.01] + [0.0] * 8]))
    with pytest.raises(PPError) as e:
        BAProblem(sc, linear_solver=1)             # the DIRECT solver's structure
    assert e.value.code == PP_ERR_INVALID and "pair entries" in str(e.value)
    BAProblem(sc).close()                          # AUTO: 47 000 images -> ITERATIVE_SCHUR, which builds no pair lists


@pytest.mark.parametrize("n,band", [(1000, 150), (2990, 300), (2990, 900)])
def test_dense_cholesky_block_sparse_input(n, band, monkeypatch):
    """a banded SPD matrix: the 64x64 tiles that are zero and stay zero in the factor get no workgroup (per-launch row / super-tile
    lists, back substitution skipping them).  Every skipped operation was a product with a zero tile, so the solution has the
    same bits as with PPSFM_CHOL_SPARSE=0."""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n + band)
    A = np.zeros((n, n))
    for i in range(0, n, 50):
        j = min(n, i + band)
        B = rng.normal(size=(j - i, 20))
        A[i:j, i:j] += B @ B.T
    A += np.diag(rng.uniform(1.0, 2.0, n)) * 20
    b = rng.normal(size=n)
    x, _ = dense_cholesky_solve(A, b, repeat=2)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
    monkeypatch.setenv("PPSFM_CHOL_SPARSE", "0")
    x1, _ = dense_cholesky_solve(A, b)                  # dense structure, paired back substitution: equal to rounding
    monkeypatch.setenv("PPSFM_BACKSUB_PAIRS", "0")
    x0, _ = dense_cholesky_solve(A, b)                  # dense structure, the block-by-block back substitution a sparse system takes: same bits
    monkeypatch.delenv("PPSFM_BACKSUB_PAIRS")
    monkeypatch.delenv("PPSFM_CHOL_SPARSE")
    assert np.array_equal(x, x0)
    assert np.allclose(x, x1, rtol=1e-11, atol=1e-13 * np.abs(x).max()) and np.linalg.norm(A @ x1 - b) / np.linalg.norm(b) < 1e-12


def _dissected_spd(rng, leaves, nsep, band, two_level=0):
    """SPD matrix in a nested-dissection order: independent banded diagonal blocks (`leaves`: their sizes), then `nsep` separator rows coupled to
    everything; two_level > 0: every PAIR of leaves is followed by a separator of that size coupled to the pair only."""
    sizes = []
    for i, l in enumerate(leaves):
        sizes.append(("leaf", l))
        if two_level and i % 2 == 1:
            sizes.append(("sep1", two_level))
    n = sum(sz for _, sz in sizes) + nsep
    A = np.zeros((n, n))
    lo, pair_lo = 0, 0
    for kind, sz in sizes:
        if kind == "leaf":
            for i in range(lo, lo + sz, 50):
                j = min(lo + sz, i + band)
                B = rng.normal(size=(j - i, 20))
                A[i:j, i:j] += B @ B.T
        else:      # rows of this separator x the columns of its pair of leaves and its own
            S1 = rng.normal(size=(sz, lo + sz - pair_lo)) * 0.3
            A[lo:lo + sz, pair_lo:lo + sz] += S1
            A[pair_lo:lo + sz, lo:lo + sz] += S1.T
            pair_lo = lo + sz
        lo += sz
    S = rng.normal(size=(nsep, n)) * 0.3
    A[lo:, :] += S
    A[:, lo:] += S.T
    A = 0.5 * (A + A.T)
    return A + np.diag(np.abs(A).sum(axis=1) + 1.0)


@pytest.mark.parametrize("shape", ["two_leaves", "uneven", "four_leaves", "two_level", "twelve_leaves"])
def test_dense_cholesky_several_chains(shape, monkeypatch):
    """A nested-dissection structure: the independent sub-trees of the elimination tree are factorised side by side, one chain workgroup per leaf
    (cholesky.hip "ChainRanges").  Same solution as with one chain (PPSFM_CHOL_CHAINS=1, the bits of the dense path) to rounding - the separators
    receive their panels in the order of the steps, not of the column index - and deterministic."""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(len(shape))
    A = {"two_leaves": lambda: _dissected_spd(rng, [1344, 1344], 300, 200),
         "uneven": lambda: _dissected_spd(rng, [576, 1920], 490, 260),
         "four_leaves": lambda: _dissected_spd(rng, [640, 640, 640, 640], 420, 150),
         "two_level": lambda: _dissected_spd(rng, [576, 576, 576, 576], 300, 150, two_level=192),
         "twelve_leaves": lambda: _dissected_spd(rng, [256] * 12, 180, 100)}[shape]()      # (more than eight chains)
    n = A.shape[0]
    b = rng.normal(size=n)
    x, ms = dense_cholesky_solve(A, b, repeat=3)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
    x2, _ = dense_cholesky_solve(A, b, repeat=2)
    assert np.array_equal(x, x2)
    monkeypatch.setenv("PPSFM_CHOL_CHAINS", "1")
    x1, ms1 = dense_cholesky_solve(A, b, repeat=3)
    monkeypatch.delenv("PPSFM_CHOL_CHAINS")
    assert np.allclose(x, x1, rtol=1e-10, atol=1e-13 * np.abs(x1).max())
    monkeypatch.setenv("PPSFM_CHOL_SPARSE", "0")
    x0, ms0 = dense_cholesky_solve(A, b, repeat=3)
    monkeypatch.delenv("PPSFM_CHOL_SPARSE")
    assert np.allclose(x, x0, rtol=1e-10, atol=1e-13 * np.abs(x0).max())
    monkeypatch.setenv("PPSFM_CHOL_MODE", "columns")      # one launch per block column over the same tile map (what a timed-out one-launch factorisation falls back to)
    xc, _ = dense_cholesky_solve(A, b)
    monkeypatch.delenv("PPSFM_CHOL_MODE")
    assert np.allclose(x, xc, rtol=1e-10, atol=1e-13 * np.abs(xc).max()) and np.linalg.norm(A @ xc - b) / np.linalg.norm(b) < 1e-12
    print("\n%s n=%d: %.3f ms with its chains, %.3f ms with one chain, %.3f ms dense" % (shape, n, ms, ms1, ms0))
    assert ms < 1.05 * ms1, "several chains must not be slower than one"


@pytest.mark.parametrize("seed", range(6))
def test_dense_cholesky_random_block_structures(seed):
    """random forests of banded leaves under random separators (the structures of tests/test_cholesky_task_order.py::test_task_plans_of_random_structures)
    as matrices: whatever chains the plan finds, the solve is the solution (residual) and repeatable bit for bit"""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    from test_cholesky_task_order import _random_forest
    rng = np.random.default_rng(2000 + seed)
    T = int(rng.integers(16, 60))
    nz = _random_forest(rng, T)
    n = 64 * T - int(rng.integers(2, 60))
    A = np.zeros((n, n))
    for i in range(T):
        for j in range(i + 1):
            if nz[i, j]:
                r0, r1, c0, c1 = 64 * i, min(n, 64 * i + 64), 64 * j, min(n, 64 * j + 64)
                blk = rng.normal(size=(r1 - r0, c1 - c0)) * 0.1
                A[r0:r1, c0:c1] = blk
    A = np.tril(A) + np.tril(A, -1).T
    A += np.diag(np.abs(A).sum(axis=1) + 1.0)
    b = rng.normal(size=n)
    x, _ = dense_cholesky_solve(A, b, repeat=1)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
    x2, _ = dense_cholesky_solve(A, b, repeat=1)
    assert np.array_equal(x, x2)


def test_banded_covisibility_block_sparse_solve_matches_oracle_and_dense_path(oracle, monkeypatch):
    """a sequence-like scene (every point seen by 6 of 24 consecutive images of 240): the reduced camera system is block-banded,
    the device skips the empty tiles in assembly, factorisation and back substitution (the reference would run Ceres'
    SPARSE_SCHUR here, bundle_adjustment.cc:275-286).  Same LM trajectory as the oracle (dense arithmetic), and bitwise the
    same parameters as the device's own dense path (PPSFM_BA_SPARSE=0)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(240, 6000, 6, seed=77, model=2, window=24)
    opts = dict(max_num_iterations=6)
    monkeypatch.setenv("PPSFM_BA_ORDERING", "band")      # (the images in the caller's - capture - order: a nested dissection sums the separators' tiles in another order, see test_sequence_scene_is_dissected_...)
    pb = BAProblem(sc)
    monkeypatch.delenv("PPSFM_BA_ORDERING")
    assert pb.structure()["chains"] == 1 and not pb.structure()["reordered"]
    s = pb.solve(ba_options(**opts))
    poses, points, _ = pb.get_parameters()
    S, rhs = pb.reduced_system(1e4)
    pb.close()
    # the structure really is banded at tile level: images more than 24 apart share no point
    assert np.all(S[6 * 60:, :6 * 30] == 0.0) and np.abs(S[6 * 40:6 * 50, 6 * 30:6 * 40]).max() > 0
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert s.num_iterations == rs.num_iterations == 6 and s.num_successful_steps == rs.num_successful_steps
    assert abs(s.final_cost - rs.final_cost) <= 1e-6 * rs.final_cost + 1e-18
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    monkeypatch.setenv("PPSFM_BA_SPARSE", "0")
    monkeypatch.setenv("PPSFM_BACKSUB_PAIRS", "0")      # (the back substitution a block-sparse system takes; the paired one of dense systems differs in the last bits)
    pd = BAProblem(sc)
    sd = pd.solve(ba_options(**opts))
    dposes, dpoints, _ = pd.get_parameters()
    Sd, rhsd = pd.reduced_system(1e4)
    pd.close()
    monkeypatch.delenv("PPSFM_BA_SPARSE")
    monkeypatch.delenv("PPSFM_BACKSUB_PAIRS")
    assert np.array_equal(poses, dposes) and np.array_equal(points, dpoints) and sd.final_cost == s.final_cost
    assert np.array_equal(np.tril(S), np.tril(Sd)) and np.array_equal(rhs, rhsd)


def test_shuffled_image_ids_get_their_banded_system_back(oracle, monkeypatch):
    """A sequence scene whose image ids are NOT in capture order (every tile of the reduced system non-zero in the caller's order):
    pp_ba_create renumbers the images internally by reverse Cuthill-McKee on the co-visibility graph - what Ceres' SPARSE_SCHUR
    ordering does for the reference (bundle_adjustment.cc:279-282) - and the block-sparse path is taken again.  Every per-image
    input / output stays in the caller's order: same solve as with the ids in capture order (mapped), as the oracle on the shuffled
    scene, and the reduced system handed out equals the one of a handle that kept the caller's order."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    base = synthetic.make_ba_scene(240, 6000, 6, seed=77, model=2, window=24)
    sc, new_of_old = synthetic.shuffle_image_ids(base, seed=5)
    opts = dict(max_num_iterations=6)
    pb = BAProblem(sc)
    st = pb.structure()
    assert st["reordered"] and st["block_sparse"] and not st["iterative"]
    assert st["nnz_natural"] == st["tiles"] and st["nnz_used"] * 2 < st["nnz_natural"]      # dense in the caller's order, banded in the handle's
    S, rhs = pb.reduced_system(1e4)
    s = pb.solve(ba_options(**opts))
    poses, points, _ = pb.get_parameters()
    assert s.linear_solver == 2
    with pytest.raises(Exception):      # a renumbered handle cannot join a point-sharded group (every rank must lay out the system alike)
        pb.set_allreduce(lambda ptr, count, op: 0, group_rank=0, group_size=2)
    pb.close()
    # (1) the caller's order kept (PP_ORDERING_NATURAL): the dense path, the same system and - to rounding - the same solve
    pn = BAProblem(sc, ordering=1)
    stn = pn.structure()
    assert not stn["reordered"] and not stn["block_sparse"] and stn["nnz_used"] == stn["tiles"]
    Sn, rhsn = pn.reduced_system(1e4)
    sn = pn.solve(ba_options(**opts))
    nposes, npoints, _ = pn.get_parameters()
    pn.close()
    # (a block whose row / column images swap roles is formed as the transpose: the same sums in another association - last-bit differences)
    assert np.allclose(S, Sn, rtol=1e-11, atol=1e-13 * np.abs(Sn).max()) and np.allclose(rhs, rhsn, rtol=1e-11, atol=1e-13 * np.abs(rhsn).max())
    assert s.num_iterations == sn.num_iterations and s.num_successful_steps == sn.num_successful_steps
    assert np.abs(poses - nposes).max() <= 1e-9 * np.abs(nposes).max() and np.abs(points - npoints).max() <= 1e-9 * np.abs(npoints).max()
    # (2) the ids in capture order: the same problem, image `old` of it is image new_of_old[old] here
    monkeypatch.setenv("PPSFM_BA_ORDERING", "band")                         # (the band order alone: no nested dissection on top of it)
    p0 = BAProblem(base)
    monkeypatch.delenv("PPSFM_BA_ORDERING")
    assert not p0.structure()["reordered"]                                # already banded: nothing to gain
    s0 = p0.solve(ba_options(**opts))
    bposes, bpoints, _ = p0.get_parameters()
    p0.close()
    assert np.abs(poses[new_of_old] - bposes).max() <= 1e-9 * np.abs(bposes).max() and np.abs(points - bpoints).max() <= 1e-9 * np.abs(bpoints).max()
    assert abs(s.final_cost - s0.final_cost) <= 1e-9 * s0.initial_cost
    # (3) the oracle on the shuffled scene (BASELINE's tolerance)
    rposes, rpoints, _, rs, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()


@pytest.mark.parametrize("shuffled,loop", [(False, False), (True, False), (True, True)])
def test_sequence_scene_is_dissected_and_factorised_by_several_chains(oracle, monkeypatch, shuffled, loop):
    """A sequence scene (block-banded reduced system): pp_ba_create orders the images by nested dissection of the band - [part | part | the images that
    couple them], parts starting at 64-column tile boundaries - and the one-launch factorisation runs a chain workgroup per part (cholesky.hip
    "ChainRanges"): fewer block-column steps on the critical path than the band has block columns.  Every per-image input / output stays in the caller's
    order; same solve as with the band order alone (PPSFM_BA_ORDERING=band: one chain) to rounding, same reduced system, the oracle's end point
    (BASELINE's tolerance), and bit-for-bit repeatable."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(330, 9000, 6, seed=91, model=2, window=12 if loop else 20, loop=loop)      # loop: the sequence closes - a ring (its band order is twice as wide)
    if shuffled:
        sc, _ = synthetic.shuffle_image_ids(sc, seed=3)
    opts = dict(max_num_iterations=6)
    pb = BAProblem(sc)
    st = pb.structure()
    assert st["reordered"] and st["block_sparse"] and st["chains"] >= 2 and st["chain_steps"] <= 0.8 * 31, st      # 1981 columns: 31 block columns
    S, rhs = pb.reduced_system(1e4)
    s = pb.solve(ba_options(**opts))
    poses, points, _ = pb.get_parameters()
    assert s.linear_solver == 2 and s.cholesky_fallbacks == 0
    pb.set_parameters(sc["poses"], sc["points"], None)
    s2 = pb.solve(ba_options(**opts))
    poses2, points2, _ = pb.get_parameters()
    assert np.array_equal(poses, poses2) and np.array_equal(points, points2) and s.final_cost == s2.final_cost
    pb.close()
    monkeypatch.setenv("PPSFM_BA_ORDERING", "band")
    p1 = BAProblem(sc)
    monkeypatch.delenv("PPSFM_BA_ORDERING")
    st1 = p1.structure()
    assert st1["chains"] == 1 and st1["block_sparse"] and st1["nnz_used"] <= st["nnz_used"], st1      # (the separators' rows fill: more tiles, fewer steps)
    S1, rhs1 = p1.reduced_system(1e4)
    s1 = p1.solve(ba_options(**opts))
    poses1, points1, _ = p1.get_parameters()
    p1.close()
    assert np.allclose(S, S1, rtol=1e-11, atol=1e-13 * np.abs(S1).max()) and np.allclose(rhs, rhs1, rtol=1e-11, atol=1e-13 * np.abs(rhs1).max())
    assert s.num_iterations == s1.num_iterations and s.num_successful_steps == s1.num_successful_steps
    assert np.abs(poses - poses1).max() <= 1e-9 * np.abs(poses1).max() and np.abs(points - points1).max() <= 1e-9 * np.abs(points1).max()
    rposes, rpoints, _, rs, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()


def test_several_chains_timeout_falls_back_to_column_launches(monkeypatch):
    """the bounded waits of the one-launch factorisation with SEVERAL chain workgroups: half of the task list missing -> every chain and task leaves,
    the host repeats the step with one launch per block column over the same (dissected) tile map and stays with that mode; same end point."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(330, 9000, 6, seed=91, model=2, window=20)
    pb = BAProblem(sc)
    assert pb.structure()["chains"] >= 2
    s0 = pb.solve(ba_options(max_num_iterations=4))
    poses0, points0, _ = pb.get_parameters()
    pb.close()
    monkeypatch.setenv("PPSFM_CHOL_TEST_DROP_TASKS", "1")
    pb = BAProblem(sc)
    s = pb.solve(ba_options(max_num_iterations=4))
    poses, points, _ = pb.get_parameters()
    pb.close()
    monkeypatch.delenv("PPSFM_CHOL_TEST_DROP_TASKS")
    assert s0.cholesky_fallbacks == 0 and s.cholesky_fallbacks == 1
    assert s.num_iterations == s0.num_iterations and s.num_successful_steps == s0.num_successful_steps
    assert np.abs(poses - poses0).max() <= 1e-9 * np.abs(poses0).max() and np.abs(points - points0).max() <= 1e-9 * np.abs(points0).max()


def test_forced_reordering_of_a_dense_scene_changes_nothing_visible(monkeypatch):
    """PPSFM_BA_ORDERING=rcm forces the internal renumbering where it does not pay (a dense co-visibility): set / get parameters,
    the reduced system and the solve are those of the caller's order (constant pose and tvec masks travel with their images)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(24, 600, 5, seed=0xC0FFEE + 41, model=2)
    sc["pose_const"][5] = 1
    sc["tvec_const_mask"][9] = 0b101
    opts = dict(max_num_iterations=4)      # (above the rounding floor of the cost: beyond it the accept / reject pattern is each association's noise)
    pn = BAProblem(sc)
    assert not pn.structure()["reordered"]
    Sn, rhsn = pn.reduced_system(100.0)
    sn = pn.solve(ba_options(**opts))
    nposes, npoints, _ = pn.get_parameters()
    pn.close()
    monkeypatch.setenv("PPSFM_BA_ORDERING", "rcm")
    pr = BAProblem(sc)
    monkeypatch.delenv("PPSFM_BA_ORDERING")
    assert pr.structure()["reordered"]
    p_in, x_in, _ = pr.get_parameters()
    assert np.array_equal(p_in, sc["poses"]) and np.array_equal(x_in, sc["points"])      # round trip through the internal order
    S, rhs = pr.reduced_system(100.0)
    s = pr.solve(ba_options(**opts))
    poses, points, _ = pr.get_parameters()
    pr.close()
    assert np.allclose(S, Sn, rtol=1e-11, atol=1e-13 * np.abs(Sn).max()) and np.allclose(rhs, rhsn, rtol=1e-11, atol=1e-13 * np.abs(rhsn).max())
    assert s.num_iterations == sn.num_iterations and s.num_successful_steps == sn.num_successful_steps
    assert np.abs(poses - nposes).max() <= 1e-9 * np.abs(nposes).max() and np.abs(points - npoints).max() <= 1e-9 * np.abs(npoints).max()
    assert np.array_equal(poses[5], sc["poses"][5]) and poses[9, 4] == sc["poses"][9, 4] and poses[9, 6] == sc["poses"][9, 6]


@pytest.mark.parametrize("loss,wide,fused,track", [(0, "0", "1", 6), (2, "0", "1", 6), (0, "1", "0", 6), (2, "1", "0", 6), (0, "1", "1", 6), (2, "1", "1", 6),
                                                   (2, "1", "1", 3), (0, "1", "1", 2)])
def test_iterative_schur_pcg_follows_the_oracle(oracle, loss, wide, fused, track, monkeypatch):
    """ITERATIVE_SCHUR + SCHUR_JACOBI (the reference's choice above 1000 images, bundle_adjustment.cc:283-286), forced on a small
    scene: the device applies the Schur complement matrix-free (ba_pcg.hip), the oracle runs the same restated Ceres CG loop on the
    explicit matrix.  Inexact steps: the LM trajectories agree iteration by iteration, and so do the conjugate-gradient counts.
    wide: the vector step of an iteration spread over many workgroups (k_pcg_wide_a / _b, two launches and a ping-pong state: the
    form a point-sharded group runs) or as ONE workgroup (k_pcg_vec, the first form), PPSFM_PCG_WIDE; fused (PPSFM_PCG_FUSED, the default): three
    launches per iteration, the product kernels take the decision and form the direction themselves - four lanes per record, eight / four / two records of a point in flight (tracks of 6 / 3 / 2)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    monkeypatch.setenv("PPSFM_PCG_WIDE", wide)
    monkeypatch.setenv("PPSFM_PCG_FUSED", fused)
    sc = synthetic.make_ba_scene(60, {6: 1500, 3: 3000, 2: 6000}[track], track, seed=0xC0FFEE + 21, model=2, window=12)
    sc["loss_type"] = loss
    sc["loss_scale"] = 0.05
    sc["tvec_const_mask"][3] = 0b010
    sc["pose_const"][7] = 1
    sc["point_const"][:11] = 1
    pb = BAProblem(sc, linear_solver=2)
    s = pb.solve(ba_options(max_num_iterations=7))
    poses, points, _ = pb.get_parameters()
    trace = pb.trace()
    pb.close()
    assert s.linear_solver == 3 and s.linear_solver_iterations > 0           # PP_LINSOLVE_PCG
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=7, iterative_schur=1))
    assert s.num_iterations == rs.num_iterations == 7 and s.num_successful_steps == rs.num_successful_steps
    assert abs(s.linear_solver_iterations - rs.linear_solver_iterations) <= 2      # (a termination test within rounding of its threshold may fall either way)
    assert np.allclose(trace[:, 0], rtrace[:, 0], rtol=1e-6, atol=1e-18)           # cost per iteration
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    assert np.array_equal(poses[7], sc["poses"][7]) and poses[3, 5] == sc["poses"][3, 5]      # constant blocks did not move


def test_iterative_schur_many_workgroup_vector_step_runs_the_same_loop(monkeypatch):
    """1500 images: forced onto the one-workgroup kernel (PPSFM_PCG_WIDE=0) the loop runs the same iterations as with the default -
    equal LM step pattern, conjugate-gradient counts within the rounding of the termination test, costs to 1e-7; the inner loops (eta = 1e-3)
    are long enough to cross the explicit-residual resets (every 10th CG iteration)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(1500, 20000, 6, seed=0xC0FFEE + 23, model=2)
    runs = []
    # the default (three launches per iteration: the product kernels take the decision and form the direction), the two-launch
    # many-workgroup vector step (PPSFM_PCG_FUSED=0: what a point-sharded group runs), the one-workgroup kernel
    for wide, fused in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("PPSFM_PCG_WIDE", wide)
        monkeypatch.setenv("PPSFM_PCG_FUSED", fused)
        pb = BAProblem(sc)
        s = pb.solve(ba_options(max_num_iterations=5, eta=1e-3))      # (the iterations above rounding level: beyond them the inner loops count noise)
        poses, points, _ = pb.get_parameters()
        runs.append((s, pb.trace().copy(), poses, points))
        pb.close()
    s1, t1, p1, x1 = runs[-1]
    for sw, tw, pw, xw in runs[:-1]:
        assert sw.linear_solver == s1.linear_solver == 3 and sw.num_iterations == s1.num_iterations
        assert sw.linear_solver_iterations > 12 * sw.num_iterations // 2      # long enough inner loops to cross the residual reset
        assert abs(sw.linear_solver_iterations - s1.linear_solver_iterations) <= 3
        big = t1[:, 0] > 1e-12 * t1[0, 0]      # (below that the accept / reject pattern is rounding)
        assert big.sum() >= 3 and np.array_equal(tw[big, 6], t1[big, 6]) and np.allclose(tw[big, 0], t1[big, 0], rtol=1e-7)
        assert np.abs(pw - p1).max() <= 1e-7 * np.abs(p1).max() and np.abs(xw - x1).max() <= 1e-7 * np.abs(x1).max()


def test_iterative_schur_iteration_cap_ends_both_vector_steps_alike(monkeypatch):
    """max_linear_solver_iterations = 3: every inner loop ends at the cap (the iterate it has is the step, as in Ceres) - the many-workgroup
    vector step and the one-workgroup kernel take that exit at the same iteration and give the same LM trajectory."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(60, 1500, 6, seed=0xC0FFEE + 24, model=2, window=12)
    runs = []
    for wide, fused in (("1", "1"), ("1", "0"), ("0", "1")):      # three launches per iteration (default) / the two-launch vector step / one workgroup
        monkeypatch.setenv("PPSFM_PCG_WIDE", wide)
        monkeypatch.setenv("PPSFM_PCG_FUSED", fused)
        pb = BAProblem(sc, linear_solver=2)
        s = pb.solve(ba_options(max_num_iterations=6, max_linear_solver_iterations=3, eta=1e-6))
        runs.append((s, pb.trace().copy()))
        pb.close()
    s1, t1 = runs[-1]
    for sw, tw in runs[:-1]:
        assert sw.num_iterations == s1.num_iterations == 6
        assert sw.linear_solver_iterations == s1.linear_solver_iterations == 3 * 6
        assert np.array_equal(tw[:, 6], t1[:, 6]) and np.allclose(tw[:, 0], t1[:, 0], rtol=1e-9, atol=1e-18) and np.allclose(tw[:, 5], t1[:, 5], rtol=1e-7)


def test_iterative_schur_is_selected_above_1000_images_and_converges_to_the_direct_solution():
    """1100 images, every point seen by 6 of them (a well-connected scene whose minimum is unique): AUTO picks the iterative solver by the
    image count exactly like BundleAdjuster::Solve (bundle_adjustment.cc:276-286); run to convergence with a tighter forcing term
    (eta = 1e-2; Ceres' default 1e-1 converges linearly and needs hundreds of LM iterations for this bar) its parameters equal those of the
    direct (dense Cholesky, 6600 columns) solve to 1e-5 - measured 7e-12."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(1100, 22000, 6, seed=0xC0FFEE + 22, model=2)
    opts = dict(max_num_iterations=60, function_tolerance=1e-16, gradient_tolerance=1e-14, parameter_tolerance=1e-14)
    pi = BAProblem(sc)
    si = pi.solve(ba_options(eta=1e-2, **opts))
    iposes, ipoints, _ = pi.get_parameters()
    pi.close()
    assert si.linear_solver == 3 and si.linear_solver_iterations > 0 and si.termination == 0
    pd = BAProblem(sc, linear_solver=1)
    sd = pd.solve(ba_options(**opts))
    dposes, dpoints, _ = pd.get_parameters()
    pd.close()
    assert sd.linear_solver in (0, 1, 2) and sd.linear_solver_iterations == 0
    assert si.final_cost <= 1e-18 * si.initial_cost and sd.final_cost <= 1e-18 * sd.initial_cost
    assert np.abs(ipoints - dpoints).max() <= 1e-5 * np.abs(dpoints).max() and np.abs(iposes - dposes).max() <= 1e-5 * np.abs(dposes).max()
    # the default forcing term: inexact steps, monotone cost, the same basin
    pj = BAProblem(sc)
    sj = pj.solve(ba_options(max_num_iterations=15))
    tr = pj.trace()
    pj.close()
    assert sj.num_successful_steps == 15 and np.all(np.diff(tr[:, 0]) < 0) and sj.final_cost <= 1e-9 * sj.initial_cost


# SIMPLE_RADIAL f + k shared by all images / one block per image; OPENCV with two blocks (everything but the principal point)
@pytest.mark.parametrize("model,nintr,const_bits,loss", [(2, 1, 0b0110, 0), (2, 20, 0b0110, 2), (4, 2, 0b00001100, 0)])
def test_iterative_schur_with_variable_intrinsics_follows_the_oracle(oracle, model, nintr, const_bits, loss):
    """ITERATIVE_SCHUR + SCHUR_JACOBI with refine_focal_length / refine_extra_params (bundle_adjustment.cc:283-286 with :490-528): the intrinsics
    columns follow the pose columns in the conjugate-gradient vectors, their part of the operator is applied from the per-observation intrinsics
    Jacobians (k_pcg_cam_t / k_pcg_cam_q), the preconditioner inverts one block per intrinsics block.  The oracle runs the restated Ceres loop on
    the explicit reduced system with the same blocks: LM trajectories, conjugate-gradient counts and parameters agree."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = _intr_scene(20, 500, 5, model, nintr, const_bits, seed=0xC0FFEE + 13 * model + nintr)
    sc["loss_type"] = loss
    sc["loss_scale"] = 0.05
    sc["pose_const"][4] = 1
    pb = BAProblem(sc, linear_solver=2)
    assert pb.structure()["iterative"]
    s = pb.solve(ba_options(max_num_iterations=8))
    poses, points, intr = pb.get_parameters()
    trace = pb.trace()
    pb.close()
    assert s.linear_solver == 3 and s.linear_solver_iterations > 0           # PP_LINSOLVE_PCG
    rposes, rpoints, rintr, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=8, iterative_schur=1))
    assert s.num_iterations == rs.num_iterations == 8 and s.num_successful_steps == rs.num_successful_steps
    assert abs(s.linear_solver_iterations - rs.linear_solver_iterations) <= 3      # (a termination test within rounding of its threshold may fall either way)
    assert np.allclose(trace[:, 0], rtrace[:, 0], rtol=1e-6, atol=1e-18)           # cost per iteration
    assert np.abs(intr - rintr).max() <= 1e-5 * np.abs(rintr).max()
    assert np.abs(points - rpoints).max() <= 1e-5 * np.abs(rpoints).max() and np.abs(poses - rposes).max() <= 1e-5 * np.abs(rposes).max()
    start = np.asarray(sc["intr"])
    for j in range(12):      # constant parameters did not move
        if (const_bits >> j) & 1:
            assert np.array_equal(intr[:, j], start[:, j])
    assert np.array_equal(poses[4], sc["poses"][4])


def test_iterative_schur_with_variable_intrinsics_takes_the_direct_solvers_steps():
    """The same problem through the direct factorisation: with the inner loop run to its end (eta = 1e-14, no cap that bites) an LM step of the
    iterative path IS the exact step, so three LM iterations land on the direct path's parameters - intrinsics included.  (Not compared at
    convergence: with free focal lengths this scene has more than one minimum.)"""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = _intr_scene(40, 1200, 6, 2, 3, 0b0110, seed=0xC0FFEE + 77)
    out = []
    for ls in (1, 2):
        pb = BAProblem(sc, linear_solver=ls)
        s = pb.solve(ba_options(max_num_iterations=3, eta=1e-14, max_linear_solver_iterations=3000))
        out.append((s, pb.get_parameters(), pb.trace().copy()))
        pb.close()
    (sd, (pd, xd, kd), td), (si, (pi, xi, ki), ti) = out
    assert sd.linear_solver != 3 and si.linear_solver == 3 and si.linear_solver_iterations > 3 * 20
    assert np.array_equal(td[:, 6], ti[:, 6]) and np.allclose(td[:, 0], ti[:, 0], rtol=1e-7)
    assert np.abs(ki - kd).max() <= 1e-7 * np.abs(kd).max() and np.abs(pi - pd).max() <= 1e-7 * np.abs(pd).max() and np.abs(xi - xd).max() <= 1e-7 * np.abs(xd).max()


def test_variable_pose_with_a_quaternion_off_unit_length_is_refused():
    """"CostFunction assumes unit quaternions" (bundle_adjustment.cc:354-355, AddImageToProblem normalises first).  The device's Jacobian on the rotation
    tangent is exact for unit q only (tools/fuzz_line_eval.py: factors of |q| otherwise), so pp_ba_set_parameters refuses a variable pose that is off unit
    length by more than 1e-6 in |q|^2 instead of taking other steps than Ceres would; a CONSTANT pose enters through the rotate-point polynomial only, as in
    the reference (AddPointToProblem does not normalise), and is accepted as given; rounding-level drift passes."""
    from privacy_preserving_sfm_amd.device import BAProblem
    from privacy_preserving_sfm_amd._capi import PPError
    sc = synthetic.make_ba_scene(6, 120, 3, seed=5)
    pb = BAProblem(sc)
    poses = np.array(sc["poses"]); poses[:, :4] /= np.linalg.norm(poses[:, :4], axis=1, keepdims=True)
    pb.set_parameters(poses, sc["points"], sc["intr"])
    drift = poses.copy(); drift[3, :4] *= 1.0 + 1e-9
    pb.set_parameters(drift, None, None)
    long = poses.copy(); long[3, :4] *= 1.01
    with pytest.raises(PPError, match="unit length"):
        pb.set_parameters(long, None, None)
    const0 = poses.copy(); const0[0, :4] *= 1.5                  # pose 0 is the gauge's constant pose of this scene
    assert sc["pose_const"][0] == 1
    pb.set_parameters(const0, None, None)
    pb.close()


@pytest.mark.parametrize("n", [8191, 8200, 9100])
def test_dense_cholesky_beyond_128_block_columns(n):
    """Above 128 block columns (8191 unknowns + the right-hand side's row) the one-launch factorisation's counter arrays end and the solve runs one launch
    per block column (cholesky.hip kMaxSteps): a direct solve forced on ~1400 images and more.  Both sides of the limit solve the system; repeatable bits."""
    from privacy_preserving_sfm_amd.device import dense_cholesky_solve
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, 64))
    A = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n)) * n
    b = rng.normal(size=n)
    x, _ = dense_cholesky_solve(A, b)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
    x2, _ = dense_cholesky_solve(A, b, repeat=2)
    assert np.array_equal(x, x2)
