"""device solve() time after a factor: python tools/solve_time.py WORKLOAD... [--nrhs N]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import baspacho_amd as B
from baspacho_amd import testing as T

nrhs = 1
names = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--nrhs":
        nrhs = int(args.pop(0))
    else:
        names.append(a)
for name in names:
    sizes, ss, ranges, desc, _ = bench.build_problem(name)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream())
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).cuda()
    sol.factor(A)
    n = sol.order()
    rhs = torch.randn(nrhs * n, dtype=torch.float64, device="cuda")
    for part, fn in (("solve", sol.solve), ("solveL", sol.solveL), ("solveLt", sol.solveLt)):
        xs = [rhs.clone() for _ in range(7)]
        torch.cuda.synchronize()
        ts = []
        for x in xs:
            t0 = time.perf_counter()
            fn(A, x, n, nrhs)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%-10s %-8s nRHS %d: %.3f ms" % (name, part, nrhs, np.median(ts[2:])))
