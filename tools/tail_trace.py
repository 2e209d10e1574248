"""where the persistent tail launch spends its time: python tools/tail_trace.py [WORKLOAD] [BLOCKS]
(sets BSP_SWEEP_TRACE=1, BSP_TAIL_BLOCKS): per spine workgroup, when it had replayed the finished
panels, when the previous diagonal block arrived, when its own was published (us from the first)."""
import ctypes
import os
import sys

os.environ["BSP_SWEEP_TRACE"] = "1"
os.environ["BSP_TAIL_BLOCKS"] = sys.argv[2] if len(sys.argv) > 2 else "6"
import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import baspacho_amd as B
from baspacho_amd import testing as T

name = sys.argv[1] if len(sys.argv) > 1 else "bal871"
sizes, ss, ranges, desc, _ = bench.build_problem(name)
sol = B.create_solver(B.Settings(), sizes, ss, ranges)
h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
sol.damp(h, 0.0, sol.order() * 1.2)
for _ in range(3):
    A = torch.from_numpy(h).cuda()
    sol.factor(A)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (4 * 4096))()
nb = ctypes.c_int32(0)
assert sol._lib.bsp_test_read_sweep_trace(sol._h, buf, 4096, ctypes.byref(nb)) == 0
t = np.array(buf[:4 * nb.value], dtype=np.int64).reshape(-1, 8)
n = int((t[:, 3] > 0).sum())
t = t[:n]
t0 = t[0, 0]
us = (t - t0) / 100.0
print("%s, tail of %s blocks: %d spines, first start -> last diagonal block %.1f us" % (name, os.environ["BSP_TAIL_BLOCKS"], n, us[:, 3].max()))
print("    q   start  potrf-at diagseen  raised |   step  hop+wait  solve+update  potrf+publish")
steps = []
for q in range(n):
    r = us[q]
    step = r[3] - us[q - 1][3] if q else float("nan")
    if q:
        steps.append((step, r[2] - us[q - 1][3], r[1] - r[2], r[3] - r[1]))
    if q < 6 or q >= n - 3 or q % 6 == 0:
        print("  %3d %7.1f %8.1f %8.1f %7.1f | %6.2f %8.2f %10.2f %12.2f | solve %.2f publish %.2f product %.2f" % (q, r[0], r[1], r[2], r[3], step, (r[2] - us[q - 1][3]) if q else float("nan"), r[1] - r[2], r[3] - r[1], r[4] - r[2], r[5] - r[4], r[1] - r[5]))
if steps:
    s = np.array(steps)
    print("  median step %.2f us = flag hop %.2f + solve and update %.2f + panel Cholesky and publish %.2f" % tuple(np.median(s, axis=0)))
