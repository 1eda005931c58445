"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU tests) used ONLY for the exchange step of a point-sharded bundle adjustment.

* independent sub-models / hypothesis batches need no collective: every rank owns its units.
* one BA across k ranks: points (with their observations) are sharded, poses replicated; the solver calls
  the callback built by `make_allreduce` once per reduction (see include/ppsfm_hip.h, pp_ba_set_allreduce).
"""
import ctypes

import numpy as np

PP_REDUCE_SUM, PP_REDUCE_MAX = 0, 1


def point_owner(num_points, world_size):
    """Owner rank of every point: round-robin (tracks have similar lengths, so observation counts balance)."""
    return np.arange(num_points, dtype=np.int64) % int(world_size)


def shard_scene_by_points(scene, rank, world_size):
    """Rank-local view of a flat BA scene: only the observations of the points this rank owns.  Pose / point /
    camera index spaces are unchanged (poses are replicated; foreign points simply have no observation)."""
    owner = point_owner(scene["points"].shape[0], world_size)
    keep = owner[scene["obs_point"]] == rank
    out = dict(scene)
    for k in ("lines", "obs_pose", "obs_point"):
        out[k] = np.ascontiguousarray(scene[k][keep])
    out["owned_points"] = np.nonzero(owner == rank)[0]
    out["ordering"] = 1      # PP_ORDERING_NATURAL: every rank of the group lays out the exchanged reduced system in the caller's image order
    return out


def group_covisibility(shard, group=None, device_type="cpu"):
    """The UNION co-visibility of a point-sharded group: every rank's own C x C byte matrix (pp_ba_covisibility, host only), element-wise MAX over the
    group - ONE all-reduce of C x C bytes at create.  Collective."""
    import torch
    import torch.distributed as dist
    from .device import covisibility
    local = covisibility(shard)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(local)
        if device_type == "cuda":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        local = t.cpu().numpy()
    return np.ascontiguousarray(local, dtype=np.uint8)


def with_group_structure(shard, union_covisibility):
    """A shard whose handle takes its image order and its tile structure from the group's union co-visibility (pp_ba_problem_desc::covisibility with
    PP_ORDERING_AUTO): every rank renumbers alike, so the group keeps the block-sparse, several-chain factorisation of the unsharded problem."""
    out = dict(shard)
    out["covisibility"] = np.ascontiguousarray(union_covisibility, dtype=np.uint8)
    out["ordering"] = 2      # PP_ORDERING_AUTO
    return out


class _DeviceArray:
    """__cuda_array_interface__ view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


def make_allreduce(group=None, device_type="cuda"):
    """Returns fn(ptr, count, op) that reduces `count` doubles in place across `group`."""
    import torch
    import torch.distributed as dist

    def fn(ptr, count, op):
        if device_type == "cuda":
            t = torch.as_tensor(_DeviceArray(ptr, count), device="cuda")
        else:
            buf = (ctypes.c_double * int(count)).from_address(int(ptr))
            t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64))      # shares the memory
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == PP_REDUCE_MAX else dist.ReduceOp.SUM, group=group)
        if device_type == "cuda":
            torch.cuda.synchronize()
        return 0
    return fn


def make_communicator(group=None, device=0):
    """RCCL communicator (device.Communicator) over the ranks of a torch.distributed group: the group's rank 0 draws the
    128-byte id, torch.distributed carries it to the others (plumbing only: the data path is the library's own collectives)."""
    import torch.distributed as dist
    from .device import Communicator
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    box = [Communicator.unique_id().tobytes() if rank == 0 else None]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    return Communicator(np.frombuffer(box[0], dtype=np.uint8), size, rank, device=device)


def submodel_layout(rank, world, submodels):
    """Process layout of `bench.py --gpus N --submodels M` (BASELINE configs[4]: M sub-models over N ranks): rank -> (sub-model,
    rank inside the sub-model's group, group size).  M must divide N; consecutive ranks share a group (neighbouring GPUs)."""
    rank, world, submodels = int(rank), int(world), int(submodels)
    if submodels < 1 or world % submodels != 0:
        raise ValueError("--submodels %d does not divide the %d ranks" % (submodels, world))
    gsize = world // submodels
    return rank // gsize, rank % gsize, gsize


def make_submodel_groups(world, submodels):
    """One torch.distributed group per sub-model.  Collective: EVERY rank of the job creates EVERY group, in the same order."""
    import torch.distributed as dist
    gsize = int(world) // int(submodels)
    return [dist.new_group(list(range(m * gsize, (m + 1) * gsize))) for m in range(int(submodels))]


def max_over_ranks(value, device_type="cuda", group=None):
    """MAX of a host scalar over the ranks (the bench's elapsed time); the tensor lives where the backend reduces (cuda: RCCL, cpu: gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device_type)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_points(points, owned, group=None):
    """After a sharded solve every rank holds the refined values of its own points: exchange them."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    full = torch.zeros(points.shape, dtype=torch.float64)
    mask = torch.zeros(points.shape[0], dtype=torch.float64)
    full[owned] = torch.from_numpy(np.ascontiguousarray(points[owned]))
    mask[owned] = 1.0
    if world > 1:
        be = dist.get_backend(group)
        if be == "nccl":
            full, mask = full.cuda(), mask.cuda()
        dist.all_reduce(full, group=group)
        dist.all_reduce(mask, group=group)
        full, mask = full.cpu(), mask.cpu()
    out = points.copy()
    got = mask.numpy() > 0
    out[got] = full.numpy()[got]
    return out
