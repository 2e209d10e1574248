"""Multi-GPU support of the batched path: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm), the symbolic plan is analysed once and broadcast, matrices of a batch are
sharded by contiguous blocks.  A single factorisation is never split across GPUs (no exchange
step in the algorithm), so there is no collective in the data path."""
import numpy as np
import torch

from . import Solver


def shard_batch(batch_size, world, rank):
    """contiguous block [begin, end) of matrix indices owned by `rank` (sizes differ by <= 1)"""
    base, extra = divmod(batch_size, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def broadcast_solver(solver, src=0, device=None, group=None):
    """Rank `src` passes its Solver, the others pass None; everyone returns a Solver with the
    same symbolic plan.  One broadcast of the length + one of the flat int64 plan."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    device = device or torch.device("cpu")
    if rank == src:
        plan = torch.from_numpy(solver.serialize_plan()).to(device)
        length = torch.tensor([plan.numel()], dtype=torch.int64, device=device)
    else:
        length = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(length, src=src, group=group)
    if rank != src:
        plan = torch.empty(int(length.item()), dtype=torch.int64, device=device)
    dist.broadcast(plan, src=src, group=group)
    if rank == src:
        return solver
    return Solver.from_plan(plan.cpu().numpy())


def plan_checksum(solver):
    """order-sensitive checksum of the skeleton (to assert that ranks agree)"""
    sk = solver.skel()
    acc = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for k in sorted(sk):
            a = sk[k].astype(np.uint64)
            w = (np.arange(len(a), dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97))
            acc = acc * np.uint64(1099511628211) + np.uint64((a * w).sum())
    return int(acc)
