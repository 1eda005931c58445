"""world_size-2 `gloo` tests (CPU) of the N>1 host path: sharding, the reduction callback the solver
calls, and additivity of the sharded quantities (checked with the oracle)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as orc
    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd.distributed import (PP_REDUCE_MAX, PP_REDUCE_SUM, gather_points, make_allreduce,
                                                        shard_scene_by_points)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn = make_allreduce(device_type="cpu")
        # the callback reduces raw memory in place
        buf = np.arange(10, dtype=np.float64) * (rank + 1)
        assert fn(buf.ctypes.data, 10, PP_REDUCE_SUM) == 0
        assert np.array_equal(buf, np.arange(10) * 3.0)
        buf = np.array([rank, -rank, 5.0])
        fn(buf.ctypes.data, 3, PP_REDUCE_MAX)
        assert np.array_equal(buf, [1.0, 0.0, 5.0])
        # sharded scene: partition of the observations; costs and normal-equation blocks add up
        sc = synthetic.make_ba_scene(8, 120, 4, seed=5, model=2)
        sh = shard_scene_by_points(sc, rank, world)
        cnt = np.array([float(len(sh["obs_pose"]))]); fn(cnt.ctypes.data, 1, PP_REDUCE_SUM)
        assert cnt[0] == len(sc["obs_pose"])
        assert np.all(sh["obs_point"] % world == rank)
        cost_local, _ = orc.ba_cost(sh)
        c = np.array([cost_local]); fn(c.ctypes.data, 1, PP_REDUCE_SUM)
        cost_full, _ = orc.ba_cost(sc)
        assert abs(c[0] - cost_full) <= 1e-12 * cost_full
        r, Jp, Jx, _ = orc.ba_eval(sh)
        U = np.zeros((8, 6, 6))
        for o in range(len(r) // 2):
            j = Jp[o].reshape(2, 6); U[sh["obs_pose"][o]] += j.T @ j
        fn(U.ctypes.data, U.size, PP_REDUCE_SUM)
        r, Jp, Jx, _ = orc.ba_eval(sc)
        Uf = np.zeros((8, 6, 6))
        for o in range(len(r) // 2):
            j = Jp[o].reshape(2, 6); Uf[sc["obs_pose"][o]] += j.T @ j
        assert np.allclose(U, Uf, rtol=1e-12, atol=1e-9)
        # every rank refines its own points; gather_points merges them
        pts = sc["points"].copy(); pts[sh["owned_points"]] += 1.0 + rank
        merged = gather_points(pts, sh["owned_points"])
        want = sc["points"].copy(); want[0::2] += 1.0; want[1::2] += 2.0
        assert np.allclose(merged, want)
        # the create-time exchange of a point-sharded group that keeps its sparsity: every rank's own co-visibility matrix, element-wise MAX
        # over the group (ONE all-reduce of C x C bytes), then the SAME image order and chain plan on every rank - and the unsharded problem's
        from privacy_preserving_sfm_amd.device import covisibility, plan_ordering
        from privacy_preserving_sfm_amd.distributed import group_covisibility, with_group_structure
        seq, _ = synthetic.shuffle_image_ids(synthetic.make_ba_scene(300, 6000, 6, seed=0xC0FFEE + 300, model=2, window=20), seed=9)
        shq = shard_scene_by_points(seq, rank, world)
        own = covisibility(shq)
        union = group_covisibility(shq)
        full = covisibility(seq)
        assert np.array_equal(union, full) and own.sum() < full.sum() and np.array_equal(union, union.T)
        oon, info = plan_ordering(with_group_structure(shq, union))
        oon_full, info_full = plan_ordering(seq)
        assert info["reordered"] and info["chains"] >= 2 and info["block_sparse"]
        assert oon.tolist() == oon_full.tolist() and info == info_full
        box = [oon.tolist() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == oon.tolist()                                  # the same order on every rank
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_sharding_and_allreduce(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_bundle_adjuster_flatten_matches_reference_setup():
    """BundleAdjuster.SetUp semantics (optim/bundle_adjustment.cc:326-542) on the host mirror, no GPU needed."""
    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd.bundle_adjustment import (BundleAdjuster, BundleAdjustmentConfig,
                                                              BundleAdjustmentOptions, Reconstruction)
    sc = synthetic.make_ba_scene(6, 40, 3, seed=9, model=4)
    rec = Reconstruction.from_scene(sc)
    cfg = BundleAdjustmentConfig()
    for i in range(4):              # images 4,5 are NOT in the problem
        cfg.AddImage(i)
    cfg.SetConstantPose(0)
    cfg.SetConstantTvec(1, [0])
    for p in range(40):
        cfg.AddVariablePoint(p)
    opt = BundleAdjustmentOptions()
    assert cfg.NumResiduals(rec) == 2 * 120
    scene, pose_index, point_index, cam_index = BundleAdjuster(opt, cfg).flatten(rec)
    # images 4 and 5 enter through AddPointToProblem as constant poses
    assert set(pose_index) == {0, 1, 2, 3, 4, 5}
    assert [int(scene["pose_const"][pose_index[i]]) for i in range(6)] == [1, 0, 0, 0, 1, 1]
    assert scene["tvec_const_mask"][pose_index[1]] == 1 and scene["tvec_const_mask"][pose_index[2]] == 0
    assert len(scene["obs_pose"]) == 120
    assert np.all(scene["camera_const_mask"] == 0xFFFF)          # refine_* all false => constant intrinsics
    assert np.all(scene["point_const"] == 0)                     # whole track inside the problem
    # without the out-of-config observers the points whose track leaves the problem become constant
    cfg2 = BundleAdjustmentConfig()
    for i in range(4):
        cfg2.AddImage(i)
    scene2, *_ = BundleAdjuster(opt, cfg2).flatten(rec)
    n_out = sum(1 for p in rec.points3D.values() if any(i >= 4 for i, _ in p.track) and any(i < 4 for i, _ in p.track))
    assert scene2["point_const"].sum() == n_out > 0
    # refine flags -> SubsetParameterization masks (OPENCV: f 0-1, pp 2-3, extra 4-7)
    opt.refine_focal_length = True
    scene3, *_ = BundleAdjuster(opt, cfg).flatten(rec)
    assert scene3["camera_const_mask"][0] == sum(1 << i for i in (2, 3, 4, 5, 6, 7))


def test_rotation_matrix_to_quaternion_round_trip():
    from privacy_preserving_sfm_amd import synthetic
    from privacy_preserving_sfm_amd.estimators import RotationMatrixToQuaternion
    rng = np.random.default_rng(0)
    for _ in range(50):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        R = synthetic.quat_to_rot(q)
        q2 = RotationMatrixToQuaternion(R)
        assert min(np.abs(q - q2).max(), np.abs(q + q2).max()) < 1e-12


def _group_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from privacy_preserving_sfm_amd import device, distributed, synthetic
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # bench.py --gpus 4 --submodels 2: two sub-models, each point-sharded over a group of two ranks
        submodels, gsize = 2, 2
        model_id, grank = rank // gsize, rank % gsize
        groups = [dist.new_group(list(range(m * gsize, (m + 1) * gsize))) for m in range(submodels)]

        class FakeCommunicator:      # stands in for the RCCL communicator (no GPU here): records what make_communicator hands it
            counter = [0]

            @staticmethod
            def unique_id():
                FakeCommunicator.counter[0] += 1
                return np.full(128, 16 * rank + FakeCommunicator.counter[0], dtype=np.uint8)

            def __init__(self, unique_id, num_ranks, rank_in_group, device=0):
                self.uid, self.size, self.rank = np.array(unique_id), num_ranks, rank_in_group

        device.Communicator = FakeCommunicator
        comm = distributed.make_communicator(groups[model_id], device=0)
        assert (comm.size, comm.rank) == (gsize, grank)
        # the id is the one drawn by the group's rank 0 (global rank model_id * gsize), on every rank of the group, and differs between groups
        assert np.all(comm.uid == 16 * (model_id * gsize) + 1)
        # sharding inside a group: the two ranks partition the sub-model's observations; the sub-models are different scenes
        sc = synthetic.make_ba_scene(8, 120, 4, seed=5 + 101 * model_id, model=2)
        sh = distributed.shard_scene_by_points(sc, grank, gsize)
        import torch
        t = torch.tensor([float(len(sh["obs_pose"]))], dtype=torch.float64)
        dist.all_reduce(t, group=groups[model_id])
        assert t.item() == len(sc["obs_pose"])
        # |x|^2 of the whole problem = pose part counted once (group rank 0) + the shards' point parts (what k_norms_partial + the sum give)
        own = sh["owned_points"]
        local = float((sc["points"][own] ** 2).sum()) + (float((sc["poses"] ** 2).sum()) if grank == 0 else 0.0)
        t = torch.tensor([local], dtype=torch.float64)
        dist.all_reduce(t, group=groups[model_id])
        assert abs(t.item() - float((sc["points"] ** 2).sum() + (sc["poses"] ** 2).sum())) <= 1e-12 * t.item()
        q.put((rank, "ok"))
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gloo_world4_two_submodels_of_two_ranks():
    """the process layout of `bench.py --gpus N --submodels M` (BASELINE configs[4]: sub-model groups with an exchange inside each)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _bench_worker(rank, world, port, argv, q):
    """bench.py's own main() on CPU: gloo instead of RCCL, a stand-in problem whose solve performs the group exchange of an LM iteration
    through the communicator - everything else (layout, groups, communicator id broadcast, sharding, timed region with barriers and
    MAX reduction, the cfg5_literal row with its watchdog thread, the JSON line) is the code the driver runs on the 8-GPU node."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import torch.distributed as dist
        import bench
        from privacy_preserving_sfm_amd import device, distributed
        bench.BA_CFG.update(num_cams=8, num_points=120, track=4)
        log = []

        class FakeCommunicator:      # stands in for the RCCL communicator: same constructor, the all-reduce goes through gloo
            pending_group = None

            @staticmethod
            def unique_id():
                return np.full(128, 7 + rank, dtype=np.uint8)

            def __init__(self, unique_id, num_ranks, rank_in_group, device=0):
                self.uid, self.size, self.rank = np.array(unique_id), num_ranks, rank_in_group
                self.group = FakeCommunicator.pending_group
                self.fn = distributed.make_allreduce(self.group, device_type="cpu")
                log.append(("comm", int(self.uid[0]), num_ranks, rank_in_group))

            def allreduce(self, ptr, count, op=0):
                assert self.fn(ptr, count, op) == 0

            def close(self):
                log.append(("comm_close",))

        device.Communicator = FakeCommunicator

        class Summary:
            pass

        class StubProblem:
            def __init__(self, scene):
                self.scene, self.M, self.comm = scene, len(scene["obs_pose"]), None
                self.C = scene["poses"].shape[0]

            def set_communicator(self, comm):
                self.comm = comm

            def set_parameters(self, poses, points, intr):
                pass

            def solve(self, o):
                k = int(o.max_num_iterations)
                for _ in range(k):
                    if self.comm is not None:      # the exchanges of one LM iteration: U/g_c, packed S, scalars
                        tot = np.array([float(self.M)]); self.comm.allreduce(tot.ctypes.data, 1, 0)      # the shards' observation counts add up to the sub-model's
                        assert tot[0] in (480.0, 176000.0, 200000.0)                                  # (the 8-camera test scene / the 1100-image scene of the iterative row / the sequence scene of the banded row)
                        u = np.full(42 * self.C, float(self.M)); self.comm.allreduce(u.ctypes.data, u.size, 0)
                        assert u[0] == tot[0]
                        spack = np.ones(6 * self.C * (6 * self.C + 1) // 2); self.comm.allreduce(spack.ctypes.data, spack.size, 0)
                        assert spack[-1] == self.comm.size
                        g = np.array([float(self.comm.rank)]); self.comm.allreduce(g.ctypes.data, 1, 1)
                        assert g[0] == self.comm.size - 1
                s = Summary()
                s.num_iterations = s.num_successful_steps = k
                s.termination, s.cholesky_fallbacks, s.linear_solver = 1, 0, 1
                return s

            def close(self):
                pass

        class StubBackend:
            dist_backend, device_type, has_gpu_rows = "gloo", "cpu", False

            def init(self, local):
                pass

            def bind_thread(self):
                pass

            def init_process_group(self):
                dist.init_process_group("gloo", rank=rank, world_size=world)

            def sync(self):
                pass

            def ba_problem(self, scene):
                return StubProblem(scene)

            def communicator(self, group):
                FakeCommunicator.pending_group = group
                return distributed.make_communicator(group, device=0)

        res = bench.main(argv, backend=StubBackend())
        q.put((rank, "ok", res, log))
    except BaseException:  # noqa  (SystemExit too)
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def _run_bench(world, argv):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg, _, _ in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
    return {rank: (r, log) for rank, _, r, log in res}


def test_bench_main_multi_rank_control_flow_under_gloo():
    """`bench.py --gpus 4` as the driver launches it (default: 4 replicas + the widened.cfg5_literal row = 2 sub-models as replicas and
    as 2 x 2 sharded ranks) and `--submodels 2` (the headline itself sharded), executed under gloo with a stand-in problem."""
    out = _run_bench(4, ["--gpus", "4", "--steps", "4", "--warmup", "1"])
    line, log0 = out[0]
    assert line["n_gpus"] == 4 and line["scaling"] == "weak" and line["config"]["submodels"] == 4 and line["config"]["ranks_per_submodel"] == 1
    assert "4 replicas of configs[2]" in line["config"]["workload"] and "widened.cfg5_literal" in line["config"]["workload"]
    row = line["widened"]["cfg5_literal"]
    assert "error" not in row and row["submodels"] == 2 and row["literal_configs4"] is False
    assert row["replicas"]["busy_gpus"] == 2 and row["sharded"]["ranks_per_submodel"] == 2 and row["sharded"]["value"] > 0
    assert row["sharded_iterative_1100"]["cams"] == 1100 and row["sharded_iterative_1100"]["value"] > 0
    assert row["sharded_banded"]["ranks_per_submodel"] == 2 and row["sharded_banded"]["value"] > 0      # (the create-time exchange of the co-visibility ran over gloo)
    assert all(out[r][0] is None for r in (1, 2, 3))                      # rank 0 alone holds the line
    # the communicator of a group carries the id drawn by the group's first rank: ranks 0,1 -> 7 + 0, ranks 2,3 -> 7 + 2
    for r in range(4):
        comms = [e for e in out[r][1] if e[0] == "comm"]      # one communicator each for the sharded row, the sharded iterative row and the sharded sequence row
        assert comms == [("comm", 7 + 2 * (r // 2), 2, r % 2)] * 3 and out[r][1].count(("comm_close",)) == 3
    out = _run_bench(4, ["--gpus", "4", "--steps", "4", "--warmup", "1", "--submodels", "2"])
    line = out[0][0]
    assert line["config"]["submodels"] == 2 and line["config"]["ranks_per_submodel"] == 2 and "RCCL" in line["config"]["exchange"]
    assert "cfg5_literal" not in line.get("widened", {})
    assert abs(line["value"] - 2 * 4 / (line["ms_per_step"] * 4e-3)) <= 1e-6 * line["value"]
