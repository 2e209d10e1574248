"""Sensitivity of factor() time to the supernode-merge cost model (ComputationModel): the fixed
per-op costs of model_Hip_MI355X scaled by a factor (larger = merge more)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import baspacho_amd as B
from baspacho_amd import testing as T

# csrc/computation_model.cpp, model_Hip_MI355X
BASE = [4.0e-07, 1.5e-07, 1.0e-10, 1.7e-14,
        3.0e-07, 2.0e-09, 0.0, 2.0e-10, 5.0e-12, 5.0e-14,
        3.0e-07, 1.0e-10, 2.0e-12, 5.0e-10, 1.0e-12, 1.0e-13,
        1.0e-07, 2.0e-09, 2.0e-09, 1.0e-10]
GROUPS = {"potrf": (0, 4), "trsm": (4, 10), "syge": (10, 16), "asmbl": (16, 20)}
FIXED = [0, 4, 10, 16]

def run(name, sizes, ss, ranges, scale, group=None):
    m = list(BASE)
    if group is None:
        for i in FIXED:
            m[i] *= scale
    else:  # scale every coefficient of one op
        for i in range(*GROUPS[group]):
            m[i] *= scale
        name = name + "/" + group
    sol = B.create_solver(B.Settings(computationModel=m), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream())
    h = T.random_data(sol.dataSize(), -1, 1, 37); sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).cuda()
    bufs = [A.clone() for _ in range(8)]
    sol.factor(bufs[0]); sol.factor(bufs[1]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 8): sol.factor(bufs[i])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    print("%-9s scale %5.2f: %8.3f ms  lumps %6d  data %.1f MB  flops %.2f GF" % (
        name, scale, ms, sol.numLumps(), sol.dataSize() * 8 / 1e6, sol.factorFlops() / 1e9), flush=True)

import bench
which = sys.argv[1] if len(sys.argv) > 1 else "fixed"
probs = {}
for name in sys.argv[2:] or ["grid82", "flat50k", "bal871"]:
    if name == "grid82":
        probs[name] = (np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37), [])
    elif name == "flat50k":
        probs[name] = (np.full(16667, 3, dtype=np.int64), T.gen_flat(16667, 3.0e-4, 37), [])
    elif name == "tridiag":
        probs[name] = (np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334), [])
    else:
        sizes, ss, ranges, _ = bench.build_problem(name)
        probs[name] = (sizes, ss, ranges)
for name, (s, ss, r) in probs.items():
    if which == "fixed":
        for scale in (0.05, 0.15, 0.5, 1.0, 2.5, 5.0, 20.0, 80.0):
            run(name, s, ss, r, scale)
    else:
        for g in GROUPS:
            for scale in (0.1, 0.3, 3.0, 10.0):
                run(name, s, ss, r, scale, g)
