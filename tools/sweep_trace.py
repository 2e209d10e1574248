"""where a persistent solve sweep spends its time: python tools/sweep_trace.py [WORKLOAD] [--nrhs N]
(needs BSP_SWEEP_TRACE=1, set here).  Prints, per spine workgroup of the LAST sweep of a solveL and of a
solveLt call: when its operands were on chip, when its inputs arrived, when it published -- relative to
the first spine's start, in microseconds -- and the step time published(b) - published(previous)."""
import ctypes
import os
import sys

os.environ["BSP_SWEEP_TRACE"] = "1"
import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import baspacho_amd as B
from baspacho_amd import testing as T

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bal871"
nrhs = int(sys.argv[sys.argv.index("--nrhs") + 1]) if "--nrhs" in sys.argv else 1
sizes, ss, ranges, desc, _ = bench.build_problem(name)
sol = B.create_solver(B.Settings(), sizes, ss, ranges)
h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
sol.damp(h, 0.0, sol.order() * 1.2)
A = torch.from_numpy(h).cuda()
sol.factor(A)
n = sol.order()
for part, fn, backward in (("solveL", sol.solveL, False), ("solveLt", sol.solveLt, True)):
    for _ in range(3):
        x = torch.randn(nrhs * n, dtype=torch.float64, device="cuda")
        fn(A, x, n, nrhs)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (4 * 4096))()
    nb = ctypes.c_int32(0)
    assert sol._lib.bsp_test_read_sweep_trace(sol._h, buf, 4096, ctypes.byref(nb)) == 0
    t = np.array(buf[:4 * nb.value], dtype=np.int64).reshape(-1, 8)
    t = t[(t[:, 3] > 0) & (t[:, 0] > 0)]
    order = np.argsort(t[:, 3])
    t = t[order]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print("%s: %d spines, first start -> last publish %.1f us" % (part, len(us), us[:, 3].max()))
    print("  blk   start  loaded  polled    done |   step  wait-after-prev  compute")
    prev = None
    steps, waits, comps = [], [], []
    for i, r in enumerate(us):
        step = r[3] - prev if prev is not None else float("nan")
        wait = r[2] - prev if prev is not None else float("nan")
        if prev is not None:
            steps.append(step); waits.append(wait); comps.append(r[3] - r[2])
        if i < 6 or i >= len(us) - 3 or i % 8 == 0:
            print("  %3d %7.2f %7.2f %7.2f %7.2f | %6.2f %8.2f %12.2f   far-in %.2f" % (order[i], r[0], r[1], r[2], r[3], step, wait, r[3] - r[2], r[4] - t0 / 100.0 * 0 if r[4] > 0 else -1))
        prev = r[3]
    if steps:
        print("  median step %.2f us = hop %.2f + compute %.2f" % (np.median(steps), np.median(waits), np.median(comps)))
