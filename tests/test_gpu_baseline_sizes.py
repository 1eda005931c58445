"""Oracle parity AT the BASELINE.json sizes (configs[1]..configs[4]).

The small-scene tests elsewhere prove the arithmetic; these prove the index structures (CSR lists, pair lists, gathers,
32-bit offsets) at the sizes the metric is quoted on, against the same CPU oracle:
  cfg 2  100 cams / 40k obs     K1 eval, reduced camera system, full LM convergence
  cfg 3  500 cams / 200k obs    K1 eval, three LM iterations (cost trace + parameters)
  cfg 4  1M P6L hypotheses over 50k correspondences: the device's best-of-H against a host arg-max over EVERY returned score
  cfg 5  shape: 2 sub-models x 2 point-sharded ranks each (threads on one GPU), results = the unsharded solves
Reference behaviour: src/optim/bundle_adjustment.cc:260-320, src/optim/ransac.h:178-278.
"""
import threading

import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_cfg2_eval_and_reduced_system_match_oracle(oracle):
    from privacy_preserving_sfm_amd.device import BAProblem
    sc = synthetic.make_ba_scene(100, 5000, 8, seed=0xC0FFEE + 2, model=2)
    pb = BAProblem(sc)
    assert pb.M == 40000
    cost, r, jp, jx, _ = pb.evaluate()
    r0, jp0, jx0, _ = oracle.ba_eval(sc)
    assert np.allclose(r, r0, rtol=1e-10, atol=1e-9)
    assert np.allclose(jp, jp0, rtol=1e-9, atol=1e-7) and np.allclose(jx, jx0, rtol=1e-9, atol=1e-7)
    assert abs(cost - 0.5 * float(r0 @ r0)) <= 1e-12 * cost
    # the damped, Jacobi-scaled 600 x 600 reduced camera system (gauge columns are identity rows)
    for radius in (1e4, 10.0):
        S, rhs = pb.reduced_system(radius)
        ref = oracle.ba_reduced_system(sc, radius)
        cols = np.array([c for c in range(600) if c >= 6 and c != 9])      # pose 0 constant, tvec[1].x constant
        assert len(cols) == ref["nc"] == 593
        scale = np.abs(ref["S"]).max()
        assert np.allclose(S[np.ix_(cols, cols)], ref["S"], rtol=1e-9, atol=1e-11 * scale)
        assert np.allclose(rhs[cols], ref["rhs"], rtol=1e-9, atol=1e-11 * np.abs(ref["rhs"]).max())
    pb.close()


def test_cfg2_solve_matches_oracle(oracle):
    """BASELINE configs[1] with full convergence: same LM trajectory (costs, accept pattern, radii), parameters <= 1e-5 rel."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(100, 5000, 8, seed=0xC0FFEE + 2, model=2)
    opts = dict(max_num_iterations=50, gradient_tolerance=1e-8)
    pb = BAProblem(sc)
    s = pb.solve(ba_options(**opts))
    poses, points, _ = pb.get_parameters()
    trace = pb.trace()
    pb.close()
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert (s.num_iterations, s.num_successful_steps, s.termination) == (rs.num_iterations, rs.num_successful_steps, rs.termination)
    k = min(len(trace), len(rtrace))
    big = rtrace[:k, 0] > 1e-14 * rtrace[0, 0]              # below that the costs are rounding noise of different summation orders
    assert big.sum() >= 4
    assert np.allclose(trace[:k, 0][big], rtrace[:k, 0][big], rtol=1e-6)
    assert np.array_equal(trace[:k, 6], rtrace[:k, 6])
    assert _rel(points, rpoints) <= 1e-5 and _rel(poses, rposes) <= 1e-5
    assert _rel(points, rpoints) <= 1e-8 and _rel(poses, rposes) <= 1e-8     # measured: far inside the contract


def test_cfg3_eval_and_three_iterations_match_oracle(oracle):
    """BASELINE configs[2] (the bench workload, 500 cams / 200k obs): K1 on all 200k observations and three LM iterations
    of the device solver against the oracle on the identical scene - cost trace <= 1e-6 rel, parameters <= 1e-5 rel."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2)
    pb = BAProblem(sc)
    assert pb.M == 200000
    cost, r, jp, jx, _ = pb.evaluate()
    r0, jp0, jx0, _ = oracle.ba_eval(sc)
    assert np.allclose(r, r0, rtol=1e-10, atol=1e-9)
    assert np.allclose(jp, jp0, rtol=1e-9, atol=1e-7) and np.allclose(jx, jx0, rtol=1e-9, atol=1e-7)
    s = pb.solve(ba_options(max_num_iterations=3))
    poses, points, _ = pb.get_parameters()
    trace = pb.trace()
    pb.close()
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=3))
    assert s.num_iterations == rs.num_iterations == 3 and s.num_successful_steps == rs.num_successful_steps
    assert len(trace) == len(rtrace) == 4
    assert np.allclose(trace[:, 0], rtrace[:, 0], rtol=1e-6)                      # costs
    assert np.allclose(trace[:, 5], rtrace[:, 5], rtol=1e-7)                      # trust-region radii (amplify the rounding of the relative decrease)
    assert np.allclose(trace[1:, 3], rtrace[1:, 3], rtol=1e-6)                    # step norms
    assert _rel(points, rpoints) <= 1e-5 and _rel(poses, rposes) <= 1e-5
    assert abs(s.final_cost - rs.final_cost) <= 1e-6 * rs.final_cost


@pytest.mark.parametrize("images,window,loop", [(500, 40, False), (1000, 40, True)])
def test_sequence_scene_at_baseline_size_dissected_matches_oracle(oracle, images, window, loop):
    """configs[2]'s size with the co-visibility of a SEQUENCE (every point inside a window of consecutive images; 1000 images: the loop closed) - the
    reference's SPARSE_SCHUR regime.  pp_ba_create dissects the image order and the one-launch factorisation runs a chain workgroup per part; three LM
    iterations against the oracle (dense arithmetic, the caller's order): cost trace <= 1e-6 rel, parameters <= 1e-5 rel."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(images, 50 * images, 8, seed=0xC0FFEE + 3, model=2, window=window, loop=loop)
    sc["linear_solver"] = 1      # (1000 images: the direct solve, as the reference still does at that count)
    pb = BAProblem(sc)
    st = pb.structure()
    assert st["block_sparse"] and st["chains"] >= 4 and st["chain_steps"] * 2 <= (6 * images + 64) // 64 + 8, st
    s = pb.solve(ba_options(max_num_iterations=3))
    poses, points, _ = pb.get_parameters()
    trace = pb.trace()
    pb.close()
    assert s.cholesky_fallbacks == 0 and s.linear_solver == 2
    rposes, rpoints, _, rs, rtrace = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(max_num_iterations=3))
    assert s.num_iterations == rs.num_iterations == 3 and s.num_successful_steps == rs.num_successful_steps
    assert np.allclose(trace[:, 0], rtrace[:, 0], rtol=1e-6)
    assert np.allclose(trace[1:, 3], rtrace[1:, 3], rtol=1e-6)
    assert _rel(points, rpoints) <= 1e-5 and _rel(poses, rposes) <= 1e-5
    assert abs(s.final_cost - rs.final_cost) <= 1e-6 * rs.final_cost


def test_cfg4_full_one_million_hypotheses():
    """BASELINE configs[3] at its full size: 1 048 576 six-tuples over 50 000 correspondences (3.9 M models, 1.9e11
    (model, correspondence) pairs - beyond 32 bits).  The device keeps only the best; here every model's (count, sum)
    is read back and the winner re-derived on the host with InlierSupportMeasurer::Compare's rule."""
    from privacy_preserving_sfm_amd.device import PoseProblem
    H, N = 1 << 20, 50000
    sc = synthetic.make_ransac_scene(N, outlier_ratio=0.5, noise_px=0.5, seed=0xBADC0DE)
    pp = PoseProblem(sc["lines"], sc["points"], sc["aligned"])
    thr = sc["max_error"] ** 2
    rep = pp.hypotheses(H, thr, seed=0)
    nm, inl, sums = pp.last_scores(H)
    assert rep.success == 1 and rep.num_trials == H and rep.hypotheses_evaluated == H
    assert nm.min() >= 0 and nm.max() <= 8
    total = int(nm.astype(np.int64).sum())
    assert rep.models_scored == total and total > 3 * H
    assert total * N > 2 ** 32                                               # the pair count does not fit 32 bits
    # host arg-max over every valid (hypothesis, model) slot: more inliers, then smaller sum, then first index
    valid = np.arange(8)[None, :] < nm[:, None]
    cnt = np.where(valid, inl.astype(np.int64), -1)
    best = cnt.max()
    cand = np.argwhere(cnt == best)
    s = sums[cand[:, 0], cand[:, 1]]
    w = cand[np.flatnonzero(s == s.min())[0]]
    assert best == rep.num_inliers and (int(w[0]), int(w[1])) == (rep.best_trial, rep.best_model_index)
    assert sums[w[0], w[1]] == rep.residual_sum
    assert (cnt[valid] <= N).all() and np.isfinite(sums[valid]).all()
    # the winner's support recomputed from scratch: the one-model scoring path and the bit-exact residuals
    model = np.array(rep.model).reshape(3, 4)
    one_inl, one_sum = pp.score(model[None], thr)
    assert one_inl[0] == rep.num_inliers and one_sum[0] == rep.residual_sum
    mask = pp.residuals(model[None])[0] <= thr
    assert mask.sum() == rep.num_inliers
    assert mask[~sc["is_outlier"]].mean() > 0.95 and mask[sc["is_outlier"]].mean() < 0.15     # vs the truth labels
    rng = np.random.default_rng(1)
    # determinism at full size
    rep2 = pp.hypotheses(H, thr, seed=0)
    assert (rep2.num_inliers, rep2.best_trial, rep2.best_model_index) == (rep.num_inliers, rep.best_trial, rep.best_model_index)
    assert list(rep2.model) == list(rep.model) and rep2.residual_sum == rep.residual_sum
    # models of sampled hypotheses (spread over the whole range: catches offset wrap-around) re-solved in a small batch score
    # the same as they did in the 1M run
    from privacy_preserving_sfm_amd.device import sampler_draw
    all_samples = sampler_draw(0, N, 6, H)
    pick = np.unique(np.concatenate([rng.integers(0, H, 30), [int(rep.best_trial), H - 1]]))
    nm2, inl2, sums2 = pp.last_scores(H)
    models, nmb = pp.p6l_batch(all_samples[pick])
    for i, h in enumerate(pick):
        assert nmb[i] == nm2[h]
        if nmb[i]:
            ci, cs = pp.score(models[i, : nmb[i]].reshape(-1, 12), thr)
            assert np.array_equal(ci, inl2[h, : nmb[i]]) and np.array_equal(cs, sums2[h, : nmb[i]])
    pp.close()


def _sharded_solve(sc, group_size, opts, errors, union_structure=False, structures=None, counts=None):
    """one BA point-sharded over `group_size` rank-threads on this GPU (pp_ba_set_allreduce); returns merged parameters.
    union_structure: the shards are created with the group's union co-visibility (every rank's own matrix, element-wise MAX - here of the rank-threads'
    pp_ba_covisibility results; tests/test_distributed_cpu.py runs the same exchange over gloo), so the group keeps the image order and the tile structure
    of the unsharded problem; structures (a list): receives every rank's pp_ba_get_structure after the attach."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options, covisibility
    from privacy_preserving_sfm_amd.distributed import _DeviceArray, shard_scene_by_points, with_group_structure
    union = None
    if union_structure:
        union = np.maximum.reduce([covisibility(shard_scene_by_points(sc, r, group_size)) for r in range(group_size)])
    barrier = threading.Barrier(group_size)
    slots = [None] * group_size
    out = [None] * group_size

    def make_fn(rank):
        def fn(ptr, count, op):
            try:
                slots[rank] = (ptr, count)
                if counts is not None and rank == 0:
                    counts.append(int(count))      # (doubles per exchange: what travels)
                barrier.wait(timeout=60)
                if rank == 0:
                    ts = [torch.as_tensor(_DeviceArray(*slots[r]), device="cuda") for r in range(group_size)]
                    res = ts[0].clone()
                    for t in ts[1:]:
                        res = torch.maximum(res, t) if op == 1 else res + t
                    for t in ts:
                        t.copy_(res)
                    torch.cuda.synchronize()
                barrier.wait(timeout=60)
                return 0
            except Exception:
                import traceback
                errors.append(traceback.format_exc())
                barrier.abort()
                return -1
        return fn

    def run(rank):
        try:
            sh = shard_scene_by_points(sc, rank, group_size)
            if union is not None:
                sh = with_group_structure(sh, union)
            pb = BAProblem(sh, ordering=sh["ordering"])
            pb.set_allreduce(make_fn(rank), group_rank=rank, group_size=group_size)
            if structures is not None:
                structures.append(pb.structure())
            s = pb.solve(ba_options(**opts))
            out[rank] = (s, pb.get_parameters(), sh["owned_points"])
            pb.close()
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(group_size)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if errors:
        return None
    poses = out[0][1][0]
    points = out[0][1][1].copy()
    for r in range(1, group_size):
        assert np.array_equal(out[r][1][0], poses)                  # replicated poses stay bitwise equal across the group
        points[out[r][2]] = out[r][1][1][out[r][2]]
    return out[0][0], poses, points


def test_cfg5_shape_two_submodels_two_ranks_each():
    """BASELINE configs[4] shape on one GPU: independent sub-models (different scenes) run concurrently, each one
    point-sharded over a group of 2 ranks that exchange the normal equations; every sub-model's
    result equals its own unsharded solve."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    torch.zeros(1, device="cuda").sum().item()
    opts = dict(max_num_iterations=8)
    scenes = [synthetic.make_ba_scene(60, 2500, 8, seed=0xC0FFEE + 5 + 101 * m, model=2) for m in range(2)]
    refs = []
    for sc in scenes:
        pb = BAProblem(sc)
        s = pb.solve(ba_options(**opts))
        refs.append((s, pb.get_parameters()))
        pb.close()
    errors = []
    results = [None, None]

    def submodel(m):
        results[m] = _sharded_solve(scenes[m], 2, opts, errors)

    th = [threading.Thread(target=submodel, args=(m,)) for m in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors[0]
    for m in range(2):
        s, poses, points = results[m]
        rs, (rposes, rpoints, _) = refs[m]
        assert s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
        assert abs(s.final_cost - rs.final_cost) <= 1e-9 * rs.initial_cost
        assert _rel(poses, rposes) <= 1e-9 and _rel(points, rpoints) <= 1e-9
    # the two sub-models are different problems (no cross-talk between groups)
    assert _rel(results[0][1], results[1][1]) > 1e-3


def test_point_sharded_banded_scene_keeps_the_block_sparse_several_chain_factorisation(oracle):
    """A sequence scene at BASELINE configs[2]'s size (500 images / 200k observations, every point inside a 40-image window, image ids shuffled),
    point-sharded over two ranks that were created with the group's UNION co-visibility (pp_ba_problem_desc::covisibility): every rank takes
    the same nested-dissection order and the same tile map, so the group factorises the exchanged system block-sparse with several chain workgroups -
    as the unsharded handle does - instead of the dense 47-step chain a PP_ORDERING_NATURAL group pays.  Merged result = the unsharded solve (1e-9), = the
    oracle (1e-5); a shard that renumbers from its OWN co-visibility is refused, by every rank of the group together."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    from privacy_preserving_sfm_amd.distributed import shard_scene_by_points
    torch.zeros(1, device="cuda").sum().item()
    sc, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 77, model=2, window=40), seed=3)
    opts = dict(max_num_iterations=3)
    pb = BAProblem(sc)
    ref_struct = pb.structure()
    rs = pb.solve(ba_options(**opts))
    rposes, rpoints, _ = pb.get_parameters()
    pb.close()
    assert ref_struct["block_sparse"] and ref_struct["chains"] >= 2
    errors, structures, counts = [], [], []
    res = _sharded_solve(sc, 2, opts, errors, union_structure=True, structures=structures, counts=counts)
    assert not errors, errors[0]
    s, poses, points = res
    # the exchanged system travels as its non-zero 64 x 64 tiles: 368 of 1128 here, 12 MB instead of the packed triangle's 36
    assert max(counts) == ref_struct["nnz_used"] * 64 * 64 and ref_struct["nnz_used"] * 3 <= ref_struct["tiles"]
    assert len(structures) == 2 and all(st == structures[0] for st in structures)                      # the same structure on every rank ...
    assert structures[0]["reordered"] and structures[0]["block_sparse"] and structures[0]["chains"] == ref_struct["chains"]      # ... the unsharded handle's
    assert structures[0]["chain_steps"] == ref_struct["chain_steps"] and structures[0]["nnz_used"] == ref_struct["nnz_used"]
    assert s.linear_solver == 2 and s.cholesky_fallbacks == 0                                            # PP_LINSOLVE_CHOLESKY_SPARSE inside the group
    assert s.num_iterations == rs.num_iterations == 3 and s.num_successful_steps == rs.num_successful_steps
    assert _rel(poses, rposes) <= 1e-9 and _rel(points, rpoints) <= 1e-9
    oposes, opoints, _, os_, _ = oracle.ba_solve(sc, oracle.BAOptionsC.defaults(**opts))
    assert _rel(poses, oposes) <= 1e-5 and _rel(points, opoints) <= 1e-5
    # the caller's order kept, the union co-visibility still given (a sequence in capture order): the group takes the union's tile map - block-sparse, one chain
    cap = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 77, model=2, window=40)
    errors2, structures2 = [], []
    from privacy_preserving_sfm_amd import distributed as _d
    keep_order = _d.with_group_structure
    _d.with_group_structure = lambda shard, union: dict(keep_order(shard, union), ordering=1)
    try:
        res2 = _sharded_solve(cap, 2, opts, errors2, union_structure=True, structures=structures2)
    finally:
        _d.with_group_structure = keep_order
    assert not errors2, errors2[0]
    assert not structures2[0]["reordered"] and structures2[0]["block_sparse"] and res2[0].linear_solver == 2
    pc = BAProblem(cap, ordering=1)
    pc.solve(ba_options(**opts))
    cposes, cpoints, _ = pc.get_parameters()
    pc.close()
    assert _rel(res2[1], cposes) <= 1e-9 and _rel(res2[2], cpoints) <= 1e-9
    # PP_ORDERING_AUTO from the shard's own observations: refused at the attach
    sh = dict(shard_scene_by_points(sc, 0, 2), ordering=2)
    pq = BAProblem(sh)
    assert pq.structure()["reordered"]
    with pytest.raises(RuntimeError, match="renumbered its images"):
        pq.set_allreduce(lambda ptr, count, op: 0, group_rank=0, group_size=1)
    pq.close()


def test_group_whose_ranks_lay_out_the_system_differently_is_refused_together():
    """Round-5 advice: the all-reduce counts of a solve follow from each rank's own tile map.  Rank 0 created with the group's union co-visibility
    (PP_ORDERING_AUTO: renumbered, block-sparse), rank 1 without it (the caller's order, dense) - the attach is collective and carries a hash of the
    layout: BOTH ranks get PP_ERR_INVALID before anything is exchanged (before: an RCCL hang or a silently corrupted system).  And pp_ba_create itself
    refuses a co-visibility matrix that lacks a pair of the rank's own shard (it cannot be the union)."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, covisibility
    from privacy_preserving_sfm_amd.distributed import _DeviceArray, shard_scene_by_points, with_group_structure
    torch.zeros(1, device="cuda").sum().item()
    sc, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(150, 3000, 6, seed=11, model=2, window=12), seed=4)
    shards = [shard_scene_by_points(sc, r, 2) for r in range(2)]
    union = np.maximum.reduce([covisibility(sh) for sh in shards])
    barrier = threading.Barrier(2)
    slots, verdicts, counts = [None, None], [None, None], []

    def fn_of(rank):
        def fn(ptr, count, op):
            slots[rank] = (ptr, count)
            counts.append(int(count))
            barrier.wait(timeout=60)
            if rank == 0:
                ts = [torch.as_tensor(_DeviceArray(*slots[r]), device="cuda") for r in range(2)]
                res = torch.maximum(ts[0], ts[1]) if op == 1 else ts[0] + ts[1]
                for t in ts:
                    t.copy_(res)
                torch.cuda.synchronize()
            barrier.wait(timeout=60)
            return 0
        return fn

    def run(rank):
        sh = with_group_structure(shards[rank], union) if rank == 0 else dict(shards[rank], ordering=1)
        pb = BAProblem(sh, ordering=sh["ordering"])
        try:
            pb.set_allreduce(fn_of(rank), group_rank=rank, group_size=2)
            verdicts[rank] = "attached"
        except RuntimeError as e:
            verdicts[rank] = str(e)
        pb.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(v is not None and "do not lay out the reduced camera system alike" in v for v in verdicts), verdicts
    assert counts == [3, 3]                                          # the one exchange of the attach; nothing else travelled
    # both with the union: attached (the same test's positive case at full size: test_point_sharded_banded_scene_keeps_...)
    # a matrix that is not the union: this shard's own pairs are missing from it
    with pytest.raises(RuntimeError, match="must be the union"):
        BAProblem(dict(shards[0], covisibility=np.zeros_like(union), ordering=2), ordering=2)


def _run_submodels_concurrently(scenes, group_size, opts):
    errors = []
    results = [None] * len(scenes)

    def submodel(m):
        results[m] = _sharded_solve(scenes[m], group_size, opts, errors)

    th = [threading.Thread(target=submodel, args=(m,)) for m in range(len(scenes))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not errors, errors[0]
    return results


def test_cfg5_full_size_four_submodels_two_ranks_each(oracle):
    """BASELINE configs[4] AT ITS OWN SIZE on one GPU: 4 independent sub-models of 500 cams / 200k obs, each point-sharded over a group
    of 2 rank-threads that exchange the normal equations per LM iteration (8 handles on one device; the 8-GPU placement itself is the
    driver's to run).  Every merged result = the sub-model's unsharded solve (1e-9), the replicated poses are bitwise equal inside a
    group (checked by _sharded_solve), one sub-model is also compared with the oracle (1e-5, BASELINE's tolerance), and the summaries
    report how many one-launch factorisations fell back.  Reference: src/controllers/incremental_mapper.h:51-55 (multiple models),
    SURVEY.md 8e rows 1-2."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    torch.zeros(1, device="cuda").sum().item()
    opts = dict(max_num_iterations=3)
    scenes = [synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 5 + 101 * m, model=2) for m in range(4)]
    refs = []
    for sc in scenes:
        assert len(sc["obs_pose"]) == 200000
        pb = BAProblem(sc)
        s = pb.solve(ba_options(**opts))
        refs.append((s, pb.get_parameters()))
        pb.close()
    results = _run_submodels_concurrently(scenes, 2, opts)
    fallbacks = 0
    for m in range(4):
        s, poses, points = results[m]
        rs, (rposes, rpoints, _) = refs[m]
        assert s.num_iterations == rs.num_iterations == 3 and s.num_successful_steps == rs.num_successful_steps
        assert abs(s.final_cost - rs.final_cost) <= 1e-9 * rs.initial_cost
        assert _rel(poses, rposes) <= 1e-9 and _rel(points, rpoints) <= 1e-9
        assert s.cholesky_fallbacks >= 0 and s.linear_solver in (0, 1)      # inside a group: the dense factorisation (one launch, or per column after a fallback)
        fallbacks += int(s.cholesky_fallbacks)
    print("cfg5 full size: cholesky_fallbacks over the 4 groups' rank-0 handles = %d" % fallbacks)
    oposes, opoints, _, os_, _ = oracle.ba_solve(scenes[0], oracle.BAOptionsC.defaults(**opts))
    assert os_.num_iterations == 3
    assert _rel(results[0][1], oposes) <= 1e-5 and _rel(results[0][2], opoints) <= 1e-5
    assert _rel(results[0][1], results[1][1]) > 1e-3                         # different problems, no cross-talk between the groups


def test_cfg5_iterative_submodel_1100_images_two_ranks():
    """The > 1000-image branch of the same shape: one 1100-image sub-model (ITERATIVE_SCHUR + SCHUR_JACOBI by the reference's rule,
    bundle_adjustment.cc:283-286) point-sharded over 2 rank-threads - 6 C doubles exchanged per CG iteration - against its unsharded
    solve (the sums arrive in another order: CG counts may differ by a few, the LM trajectory agrees to rounding)."""
    import torch
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    torch.zeros(1, device="cuda").sum().item()
    sc = synthetic.make_ba_scene(1100, 22000, 8, seed=0xC0FFEE + 5, model=2)
    opts = dict(max_num_iterations=3)
    pb = BAProblem(sc)
    rs = pb.solve(ba_options(**opts))
    rposes, rpoints, _ = pb.get_parameters()
    pb.close()
    assert rs.linear_solver == 3
    (s, poses, points), = _run_submodels_concurrently([sc], 2, opts)
    assert s.linear_solver == 3 and s.num_iterations == rs.num_iterations and s.num_successful_steps == rs.num_successful_steps
    assert abs(s.linear_solver_iterations - rs.linear_solver_iterations) <= 3
    assert abs(s.final_cost - rs.final_cost) <= 1e-6 * rs.final_cost + 1e-18
    assert _rel(poses, rposes) <= 1e-7 and _rel(points, rpoints) <= 1e-7


@pytest.mark.parametrize("window", [None, 40, -40])
def test_pair_lists_built_on_the_device_equal_the_host_builders(monkeypatch, window):
    """pp_ba_create builds the Schur pair lists of a large problem on the device (csrc/pair_lists.hip: atomic list lengths and fill positions, then a per-list
    sort that makes the result independent of the atomics' order) and of a small one - or with PPSFM_BA_PAIR_LISTS=host - on the host: the same lists, so the
    same reduced system and the same solve bit for bit.  BASELINE configs[2]'s size: the dense scene (125k lists of ~6 entries, `k_schur_blocks`) and the
    sequence scene (lists of ~35 entries in chunks of 16, image order renumbered)."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(500, 25000, 8, seed=0xC0FFEE + 3, model=2, window=abs(window) if window else None)
    if window is not None and window < 0:
        # image ids shuffled, every 17th image and every 29th point constant: the co-visibility graph the order is chosen on (built on the device beside the
        # lists: CoVisibilityOnDevice) leaves the same images and tracks out as the host's walk over the observations
        sc, _ = synthetic.shuffle_image_ids(sc, seed=11)
        sc["pose_const"] = np.ascontiguousarray(sc["pose_const"]).copy(); sc["pose_const"][::17] = 1
        sc["point_const"] = np.ascontiguousarray(sc["point_const"]).copy(); sc["point_const"][::29] = 1
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("PPSFM_BA_PAIR_LISTS", mode)
        pb = BAProblem(sc)
        S, rhs = pb.reduced_system(1e4)
        s = pb.solve(ba_options(max_num_iterations=3))
        out[mode] = (S, rhs, pb.get_parameters(), s.final_cost, pb.structure())
        pb.close()
    monkeypatch.delenv("PPSFM_BA_PAIR_LISTS")
    (S, rhs, (poses, points, _), cost, st), (S2, rhs2, (poses2, points2, _), cost2, st2) = out["device"], out["host"]
    assert st == st2 and (window is None or st["reordered"])
    assert np.array_equal(S, S2) and np.array_equal(rhs, rhs2)
    assert np.array_equal(poses, poses2) and np.array_equal(points, points2) and cost == cost2


@pytest.mark.parametrize("points,fallback", [(5000, False), (12000, True)])
def test_long_pair_lists_on_the_device_and_the_fall_back_to_the_host_builder(monkeypatch, points, fallback):
    """Few images, many shared points (12 images, every point seen by 6): 66 lists of ~1100 entries (5000 points: the device builder ranks every entry
    inside its list, k_pl_rank) or ~2700 (12 000 points: longer than the device builder takes, kMaxSortedList = 2048 - it counts the lists, finds that
    out and hands the structure to the host builder).  The same handle as with the host builder asked for outright, bit for bit."""
    from privacy_preserving_sfm_amd.device import BAProblem, ba_options
    sc = synthetic.make_ba_scene(12, points, 6, seed=0xC0FFEE + 71, model=2)
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("PPSFM_BA_PAIR_LISTS", mode)
        pb = BAProblem(sc)
        S, rhs = pb.reduced_system(1e4)
        s = pb.solve(ba_options(max_num_iterations=3))
        out[mode] = (S, rhs, pb.get_parameters(), s.final_cost)
        pb.close()
    monkeypatch.delenv("PPSFM_BA_PAIR_LISTS")
    (S, rhs, (poses, points_, _), cost), (S2, rhs2, (poses2, points2, _), cost2) = out["device"], out["host"]
    assert np.array_equal(S, S2) and np.array_equal(rhs, rhs2)
    assert np.array_equal(poses, poses2) and np.array_equal(points_, points2) and cost == cost2
