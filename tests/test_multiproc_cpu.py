"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- symbolic plan analysed on rank 0,
broadcast, rebuilt on the other rank; batch sharding."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from baspacho_amd.distributed import broadcast_solver, plan_checksum, shard_batch
    sol = None
    if rank == 0:
        from helpers import solver_random
        sol, _, _ = solver_random(57, fill=0.03, elim=(0, 60), ranges=[0, 60])
    sol = broadcast_solver(sol, src=0)
    mine = torch.tensor([plan_checksum(sol) % (2 ** 62), sol.dataSize(), sol.order(),
                         len(sol.sparseEliminationRanges())], dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok = all(torch.equal(g, gathered[0]) for g in gathered)
    b, e = shard_batch(7, world, rank)
    sizes = torch.tensor([e - b], dtype=torch.int64)
    dist.all_reduce(sizes)
    out.put((rank, ok, int(sizes.item()), sol.planStats()["num_panels"]))
    dist.barrier()
    dist.destroy_process_group()


def test_plan_broadcast_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] == 7 for r in res), res
    assert res[0][3] == res[1][3] > 0


def test_shard_batch_partitions():
    from baspacho_amd.distributed import shard_batch
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_runner_shards_a_batch_world2_metadata():
    """the shapes bench.py --gpus N uses: a batch of 64 over 1/2/4/8 ranks is 64/N each, and the
    strong-scaling value counts every matrix once"""
    from baspacho_amd.distributed import shard_batch
    for world in (1, 2, 4, 8):
        assert [shard_batch(64, world, r)[1] - shard_batch(64, world, r)[0] for r in range(world)] == \
            [64 // world] * world
