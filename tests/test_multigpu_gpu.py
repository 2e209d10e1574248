"""2-rank RCCL test of the N>1 path (skips unless two GPUs are visible): rank 0 analyses, the plan
is broadcast over RCCL, each rank factors its shard of a batch on its own GPU and checks it against
the dense Cholesky; no collective inside a factorisation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from baspacho_amd.distributed import broadcast_solver, shard_batch
    from helpers import solver_random, spd_data, dense_lower_chol, lower_of
    sol = None
    if rank == 0:
        sol, _, _ = solver_random(57, fill=0.03, elim=(0, 60), ranges=[0, 60])
    sol = broadcast_solver(sol, src=0, device=dev)
    b, e = shard_batch(6, world, rank)
    datas = [spd_data(sol, 30 + q) for q in range(b, e)]
    devs = [torch.from_numpy(d).to(dev) for d in datas]
    sol.factor(devs)          # batched factor of this rank's shard
    torch.cuda.synchronize(dev)
    worst = 0.0
    for d, t in zip(datas, devs):
        L, _ = dense_lower_chol(sol, d)
        worst = max(worst, float(np.linalg.norm(lower_of(sol, t.cpu().numpy()) - L) / np.linalg.norm(L)))
    out.put((rank, e - b, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharded_over_two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[1] for r in res) == 6
    assert all(r[2] < 1e-10 for r in res), res
