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


def broadcast_solver(solver, src=0, device=None, group=None, stats=None):
    """Rank `src` passes its Solver, the others pass None; everyone returns a Solver with the
    same symbolic plan.  One broadcast of the length + one of the flat int64 plan.
    stats (optional dict): filled with the plan's size and this rank's wall times -- `serialize_s`
    (src only), `broadcast_s` (both broadcasts, device-synchronised), `rebuild_s` (Solver.from_plan
    on the receiving ranks: symbolic contexts + device plan)."""
    import time
    import torch.distributed as dist
    rank = dist.get_rank(group)
    device = device or torch.device("cpu")
    on_gpu = torch.device(device).type == "cuda"
    t0 = time.perf_counter()
    if rank == src:
        plan = torch.from_numpy(solver.serialize_plan()).to(device)
        length = torch.tensor([plan.numel()], dtype=torch.int64, device=device)
    else:
        length = torch.zeros(1, dtype=torch.int64, device=device)
    if on_gpu:
        torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    dist.broadcast(length, src=src, group=group)
    if rank != src:
        plan = torch.empty(int(length.item()), dtype=torch.int64, device=device)
    dist.broadcast(plan, src=src, group=group)
    if on_gpu:
        torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    out = solver if rank == src else Solver.from_plan(plan.cpu().numpy())
    t3 = time.perf_counter()
    if stats is not None:
        stats.update({"plan_int64_words": int(plan.numel()), "plan_MB": round(plan.numel() * 8 / 1e6, 3),
                      "serialize_s": round(t1 - t0, 4) if rank == src else 0.0,
                      "broadcast_s": round(t2 - t1, 4),
                      "rebuild_s": round(t3 - t2, 4) if rank != src else 0.0})
    return out


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
