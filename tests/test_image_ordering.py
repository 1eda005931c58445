"""The image order pp_ba_create gives the reduced camera system - reverse Cuthill-McKee, then a nested dissection of the band whose parts the one-launch
factorisation runs side by side (csrc/ba_eval.hip: ReverseCuthillMcKee, DissectBand; csrc/cholesky.hip: PlanChains) - computed on the host alone through
pp_ba_plan_ordering.  What Ceres' SPARSE_SCHUR ordering does for the reference between 50 and 1000 images (src/optim/bundle_adjustment.cc:279-282); the
numerics of the orders are covered on the GPU (tests/test_gpu_bundle_adjustment.py::test_sequence_scene_*)."""
import numpy as np
import pytest

from privacy_preserving_sfm_amd import synthetic
from privacy_preserving_sfm_amd.device import plan_ordering


def _scene(images, window, loop=False, track=6, **kw):
    return synthetic.make_ba_scene(images, 20 * images, track, seed=0xC0FFEE + images, model=2, window=window, loop=loop, **kw)


def _covisible(sc):
    C = sc["poses"].shape[0]
    A = np.zeros((C, C), dtype=bool)
    order = np.argsort(sc["obs_point"], kind="stable")
    pts, poses = sc["obs_point"][order], sc["obs_pose"][order]
    start = np.flatnonzero(np.r_[True, pts[1:] != pts[:-1], True])
    for a, b in zip(start[:-1], start[1:]):
        v = poses[a:b]
        A[np.ix_(v, v)] = True
    return A


@pytest.mark.parametrize("images,window,loop,shuffle", [(300, 20, False, False), (500, 40, False, True), (500, 40, True, False), (1000, 10, False, True)])
def test_sequence_scenes_are_dissected_into_independent_parts(images, window, loop, shuffle):
    sc = _scene(images, window, loop)
    if shuffle:
        sc, _ = synthetic.shuffle_image_ids(sc, seed=7)
    oon, info = plan_ordering(sc)
    T = info["block_columns"]
    assert sorted(oon.tolist()) == list(range(images))                          # a permutation of the images
    assert info["reordered"] and info["block_sparse"] and info["chains"] >= 2 and info["chain_steps"] <= 0.7 * T, info
    # the structure the chains rely on: in the internal order the tile map has `chains` diagonal blocks that no tile couples with each other, each
    # starting at a tile boundary - checked on the co-visibility itself: the block columns before the first chain start of the second part share no point
    # with the second part's first block columns
    A = _covisible(sc)[np.ix_(oon, oon)]
    img_tile = (6 * np.arange(images)) // 64                                   # tile of an image's first column
    tiles = np.zeros((T, T), dtype=bool)
    ii, jj = np.nonzero(A)
    for di in (0, 5):
        for dj in (0, 5):
            tiles[(6 * ii + di) // 64, (6 * jj + dj) // 64] = True
    tiles = np.tril(tiles | tiles.T)
    starts = [k for k in range(3, T - 3) if not tiles[k:k + 3, :k].any()]
    assert len(starts) >= info["chains"] - 1, (starts, info)
    # natural ordering requested: nothing moves
    oon1, info1 = plan_ordering(sc, ordering=1)
    assert not info1["reordered"] and oon1.tolist() == list(range(images)) and info1["chains"] == 1


def test_dense_scene_keeps_the_callers_order():
    sc = synthetic.make_ba_scene(120, 3000, 8, seed=5, model=2)
    oon, info = plan_ordering(sc)
    assert not info["reordered"] and oon.tolist() == list(range(120)) and info["chains"] == 1 and not info["block_sparse"]
    assert info["chain_steps"] == info["block_columns"] == (6 * 120 + 1 + 63) // 64


def test_variable_intrinsics_stay_behind_the_pose_columns():
    sc = _scene(300, 20, num_intrinsics=1)
    sc["camera_const_mask"] = np.array([0b0110], dtype=np.uint16)
    oon, info = plan_ordering(sc)
    assert info["intrinsics_columns"] == 2 and info["reordered"] and info["chains"] >= 2 and info["block_sparse"]
    oon0, info0 = plan_ordering(_scene(300, 20))
    assert oon.tolist() == oon0.tolist()                                           # the same dissection as without them


def test_more_than_a_thousand_images_are_left_to_the_iterative_solver():
    sc = _scene(1100, 20, track=4)
    oon, info = plan_ordering(sc)                                                  # AUTO: ITERATIVE_SCHUR above 1000 images - no reduced system, no ordering
    assert not info["reordered"] and info["nnz_natural"] == -1
    oon, info = plan_ordering(sc, linear_solver=1)                                 # the direct solve requested
    assert info["reordered"] and info["chains"] >= 8


def _islands(A, oon, T, min_start=3):
    """block columns (tile indices) at which a chain can start in the internal order: rows k..k+2 of the tile map hold nothing left of column k"""
    images = A.shape[0]
    B = A[np.ix_(oon, oon)]
    tiles = np.zeros((T, T), dtype=bool)
    ii, jj = np.nonzero(B)
    for di in (0, 5):
        for dj in (0, 5):
            tiles[(6 * ii + di) // 64, (6 * jj + dj) // 64] = True
    tiles = np.tril(tiles | tiles.T)
    return [k for k in range(min_start, T - 3) if not tiles[k:k + 3, :k].any()]


@pytest.mark.parametrize("topology,clusters", [("star", 5), ("chain", 4), ("star", 8)])
def test_clustered_collection_is_dissected_by_separators_of_the_graph_itself(topology, clusters):
    """Photo collections: dense groups of images joined by a few bridge images (a hub with satellites / a chain of groups), ids shuffled.  The plain
    Cuthill-McKee band of such a graph is ONE chain over all block columns; a vertex separator - the bridge images - leaves the groups as independent parts,
    each a chain of the one-launch factorisation.  Two sources of candidates find such separators: the cuts of the band whose parts are re-ordered by their own
    Cuthill-McKee (round 4; on these scenes it finds the same cuts) and the level-structure separators of the graph itself (round 5: DissectGraph, the
    components a separator leaves as parts of their own); the fewest chain steps win.  What Ceres' SPARSE_SCHUR ordering handles for the reference on any graph
    (src/optim/bundle_adjustment.cc:279-282)."""
    sc = synthetic.make_ba_scene(500, 10000, 6, seed=0xC0FFEE + 11 * clusters, model=2, clusters=clusters, bridge=4, topology=topology)
    sc, _ = synthetic.shuffle_image_ids(sc, seed=5)
    oon, info = plan_ordering(sc)
    T = info["block_columns"]
    assert sorted(oon.tolist()) == list(range(500))
    assert info["reordered"] and info["block_sparse"] and info["chains"] >= 2, info
    # >= 1.5 x fewer chain steps than one chain over all block columns
    assert info["chain_steps"] * 3 <= T * 2, info
    assert len(_islands(_covisible(sc), oon, T)) >= info["chains"] - 1
    if topology == "star":      # the band order alone (no dissection): one chain
        import os
        os.environ["PPSFM_BA_ORDERING"] = "band"
        try:
            _, band_info = plan_ordering(sc)
        finally:
            del os.environ["PPSFM_BA_ORDERING"]
        assert band_info["chain_steps"] > info["chain_steps"], (band_info, info)


def test_per_image_intrinsics_sit_beside_their_pose_columns(monkeypatch):
    """Every image with its own camera, f and k variable (refine_focal_length / refine_extra_params, src/optim/bundle_adjustment.cc:490-528): the intrinsics
    columns couple with the same images as their image's pose, so beside the pose columns (8 columns per image) they belong to its part of the dissection;
    behind all pose columns (PPSFM_BA_INTR_LAYOUT=tail) they are dense block rows at the end of every chain.  A shared camera keeps the tail."""
    sc = _scene(500, 40, num_intrinsics=500)
    sc["camera_const_mask"] = np.full(500, 0b0110, dtype=np.uint16)
    oon, info = plan_ordering(sc)
    monkeypatch.setenv("PPSFM_BA_INTR_LAYOUT", "tail")
    _, tail = plan_ordering(sc)
    monkeypatch.delenv("PPSFM_BA_INTR_LAYOUT")
    assert info["intrinsics_columns"] == tail["intrinsics_columns"] == 1000 and info["block_columns"] == tail["block_columns"] == (8 * 500 + 1 + 63) // 64
    assert info["chains"] >= 2 and info["chain_steps"] * 4 <= tail["chain_steps"] * 3 and info["nnz_used"] * 3 <= tail["nnz_used"] * 2, (info, tail)
    odd = dict(sc, camera_const_mask=np.full(500, 0b1110, dtype=np.uint16))      # one variable parameter per camera: 7 columns per image would misalign the pose blocks - the tail
    _, info_odd = plan_ordering(odd)
    monkeypatch.setenv("PPSFM_BA_INTR_LAYOUT", "tail")
    _, tail_odd = plan_ordering(odd)
    monkeypatch.delenv("PPSFM_BA_INTR_LAYOUT")
    assert info_odd == tail_odd


def test_covisibility_matrix_and_orders_from_it():
    """pp_ba_covisibility = the image pairs that share a variable point (constant poses / points leave no edge); an order planned from the matrix passed in the
    descriptor (pp_ba_problem_desc::covisibility - the union over the shards of a point-sharded group) equals the order planned from the observations."""
    from privacy_preserving_sfm_amd.device import covisibility
    sc = _scene(300, 20)
    sc, _ = synthetic.shuffle_image_ids(sc, seed=4)
    A = _covisible(sc)
    np.fill_diagonal(A, False)
    fixed = np.flatnonzero(sc["pose_const"])
    A[fixed, :] = False; A[:, fixed] = False
    M = covisibility(sc)
    assert M.dtype == np.uint8 and np.array_equal(M.astype(bool), A) and np.array_equal(M, M.T)
    sc2 = dict(sc, point_const=np.ones(len(sc["points"]), dtype=np.uint8))
    assert not covisibility(sc2).any()                                             # constant points couple nothing
    oon, info = plan_ordering(sc)
    # half of the observations + the full matrix: the same order and plan (what every rank of a group computes)
    half = dict(sc)
    keep = sc["obs_point"] % 2 == 0
    for k in ("lines", "obs_pose", "obs_point"):
        half[k] = np.ascontiguousarray(sc[k][keep])
    oon_h, info_h = plan_ordering(dict(half, covisibility=M))
    assert oon_h.tolist() == oon.tolist() and info_h == info
    oon_own, info_own = plan_ordering(half)                                        # from its own half: another graph (fewer edges), in general another order
    assert covisibility(half).sum() < M.sum()


def test_host_pair_list_builder_against_a_brute_force_walk():
    """pp_ba_pair_lists_host (the library's host builder of the Schur pair lists, csrc/pair_lists.hip BuildPairListsOnHost - the builder of small problems and
    the one the device-built lists are compared with): on sequence / dense scenes with constant images and points, observations grouped by point and in random
    order, on 1 / 3 / 8 threads - the lists of a brute-force walk over the tracks: every pair of variable images (ci >= cj) that share a variable point, the
    (observation of ci, observation of cj) pairs in (oi, oj) order, lists in (ci, cj) order."""
    import ctypes as C
    from privacy_preserving_sfm_amd import _capi, synthetic
    from privacy_preserving_sfm_amd.device import _ba_desc
    L = _capi.lib()
    rng = np.random.default_rng(5)
    for case, (Cn, P, track, kw) in enumerate(((40, 600, 5, dict(window=10)), (16, 300, 6, {}), (9, 40, 9, {}))):
        sc = synthetic.make_ba_scene(Cn, P, track, seed=31 + case, model=2, **kw)
        pc = np.zeros(Cn, dtype=np.uint8); pc[rng.choice(Cn, 3, replace=False)] = 1
        qc = np.zeros(P, dtype=np.uint8); qc[rng.random(P) < 0.1] = 1
        sc = dict(sc, pose_const=pc, point_const=qc)
        if case == 1:      # observations in random order: the lists are sorted, not merely walked
            perm = rng.permutation(len(sc["obs_pose"]))
            sc = dict(sc, obs_pose=np.asarray(sc["obs_pose"])[perm], obs_point=np.asarray(sc["obs_point"])[perm], lines=np.asarray(sc["lines"])[perm])
        op, oq = np.asarray(sc["obs_pose"]), np.asarray(sc["obs_point"])
        want = {}
        for p in range(P):
            if qc[p]:
                continue
            obs = np.nonzero(oq == p)[0]
            for oi in obs:
                for oj in obs:
                    ci, cj = int(op[oi]), int(op[oj])
                    if oi == oj or pc[ci] or pc[cj] or cj > ci:
                        continue
                    if ci == cj and oj > oi:      # (one image seeing a point twice lists both orders once each: the walk's (f == e) rule drops only the self term)
                        pass
                    want.setdefault((ci, cj), []).append((int(oi), int(oj)))
        keys = sorted(want)
        ref_entries = [e for k in keys for e in sorted(want[k])]
        for threads in (1, 3, 8):
            keep = []
            d = _ba_desc(sc, keep)
            nl, ne = C.c_int64(0), C.c_int64(0)
            assert L.pp_ba_pair_lists_host(C.byref(d), threads, C.byref(nl), C.byref(ne), None, None, None, 0, 0) == 0
            ps = np.zeros(nl.value + 1, dtype=np.int32); pij = np.zeros((nl.value, 2), dtype=np.int32); pe = np.zeros((ne.value, 2), dtype=np.int32)
            assert L.pp_ba_pair_lists_host(C.byref(d), threads, C.byref(nl), C.byref(ne), ps.ctypes.data_as(_capi.c_ip), pij.ctypes.data_as(_capi.c_ip),
                                           pe.ctypes.data_as(_capi.c_ip), nl.value, ne.value) == 0
            assert [tuple(r) for r in pij] == keys, (case, threads)
            assert [tuple(r) for r in pe] == ref_entries, (case, threads)
            assert list(np.diff(ps)) == [len(want[k]) for k in keys]
