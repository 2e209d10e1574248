"""Plan-level tuning of the supernode merges for the batch size (round 6, verdict item 2).
The merge test of the level-scheduled backend (csrc/elimination_tree.cpp computeMerges, csrc/solver.cpp):
    merge child into parent  <=>  batch x (throughput terms of the ops, merged - separate)  <  levelCost x levels saved
Sweeps levelCost (merge aggressiveness) for a factor() of 1 / 8 / 64 matrices of GRID 82x82, with the model
planning for one matrix and for the actual batch; prints levels, lumps, flops and ms.
usage: python tools/model_batch_sweep.py <model_fit.json> [workload]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import baspacho_amd as B
from baspacho_amd import testing as T

fit = json.load(open(sys.argv[1]))
workload = sys.argv[2] if len(sys.argv) > 2 else "grid82"
def model_with(share):
    """the fitted model with every op's constant term scaled by `share` (0: a level is one launch, an
    op's own fixed cost is not paid per lump; 1: the per-op samples as they are -- merges everything
    that saves a launch of the per-op driver)"""
    m = []
    for k in ("potrf", "trsm", "syge", "asmbl"):
        p = list(fit[k]["params"])
        p[0] *= share
        m += p
    return m
sizes, ss, ranges, desc, _ = bench.build_problem(workload)
print("# %s; model = %s with the constant terms dropped" % (desc, sys.argv[1]))
print("# batch  plans-for  levelCost_us  share |  lumps levels  GF/matrix |   ms/call   ms/matrix")
CASES = [(1, lc, 0.0) for lc in (3.5, 28, 224)] + [(1, 28, sh) for sh in (0.003, 0.01, 0.03, 0.1, 0.3, 1.0)]
for batch in (1, 8, 64):
    cases = list(CASES) + ([(batch, 28, sh) for sh in (0.0, 0.03, 0.3)] if batch > 1 else [])
    cases.append((1, None, None))   # the library's built-in model and defaults
    for plan_for, lc, share in cases:
        if True:
            if share is None:
                st = B.Settings()
                lc, share = float("nan"), float("nan")
            else:
                st = B.Settings(computationModel=model_with(share), hipOptions={"expected_batch": plan_for, "level_cost_us": lc})
            sol = B.create_solver(st, sizes, ss, ranges)
            sol.setStream(torch.cuda.current_stream())
            h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
            sol.damp(h, 0.0, sol.order() * 1.3)
            base = torch.from_numpy(h).cuda()
            reps = 6
            bufs = [[base.clone() for _ in range(batch)] for _ in range(reps + 2)]
            arg = (lambda q: bufs[q] if batch > 1 else bufs[q][0])
            sol.factor(arg(0)); sol.factor(arg(1)); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for q in range(2, reps + 2):
                sol.factor(arg(q))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            stt = sol.planStats()
            print("  %5d  %8d  %11.1f  %5.3f | %6d %6d  %9.3f | %9.3f  %9.4f" % (
                batch, plan_for, lc, share, sol.numLumps(), stt["num_levels"], sol.factorFlops() / 1e9, ms, ms / batch), flush=True)
            del bufs, sol
            torch.cuda.empty_cache()
