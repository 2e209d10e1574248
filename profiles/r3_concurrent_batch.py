"""experiment: a small batch of identical-structure matrices as G concurrent sub-batches (G Solver
clones from the serialized plan, one torch stream each) against ONE batched factor() call"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import baspacho_amd as B
from baspacho_amd import testing as T

dev = torch.device("cuda", 0)
sol = B.create_solver(B.Settings(), np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37))
plan = sol.serialize_plan()
GMAX = 8
clones = [sol] + [B.Solver.from_plan(plan) for _ in range(GMAX - 1)]
streams = [torch.cuda.Stream(device=dev) for _ in range(GMAX)]
for s, st in zip(clones, streams):
    s.setStream(st)
hosts = []
for q in range(64):
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37 + q)
    sol.damp(h, 0.0, sol.order() * 1.3)
    hosts.append(torch.from_numpy(h).to(dev))

def run(batch, G, reps=5):
    mats = hosts[:batch]
    groups = [mats[g::G] for g in range(G)]
    ts = []
    for rep in range(reps + 2):
        bufs = [[a.clone() for a in grp] for grp in groups]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in range(G):
            if bufs[g]:
                clones[g].factor(bufs[g] if len(bufs[g]) > 1 else bufs[g][0])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:]))

for batch in (1, 2, 4, 8, 16, 32, 64):
    row = ["batch %2d:" % batch]
    for G in (1, 2, 4, 8):
        if G <= batch:
            row.append("G=%d %.3f ms" % (G, run(batch, G)))
    print("  ".join(row), flush=True)
