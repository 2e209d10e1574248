"""Per chain-step launch: when its first / last workgroup starts and its last workgroup ends, against
the end of workgroup 0 (the next potrf) -- library built with BSP_KTRACE=1 -DBSP_TRACE_TILE=1."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import baspacho_amd as bsp
from baspacho_amd import testing as T

dev = torch.device("cuda", 0)
sizes, ss, cam, pt = T.gen_bal_synthetic()
sol = bsp.create_solver(bsp.Settings(), sizes, ss, [0, 527480])
sol.setStream(torch.cuda.current_stream(dev))
h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
sol.damp(h, 0.0, sol.order() * 1.2)
A = torch.from_numpy(h).to(dev)
bsp.debug_read_extents()
for it in range(3):
    buf = A.clone()
    torch.cuda.synchronize()
    sol.factor(buf)
    torch.cuda.synchronize()
    ex = bsp.debug_read_extents().astype(np.float64)
ex = ex[(ex[:, 2] > 0) & (ex[:, 0] < 1e18)]
ex = ex[np.argsort(ex[:, 0])]
t0 = ex[0, 0]
print("launches", len(ex), "(10 ns units)")
print("   #   start  | last-start  wg0-end  last-end | next launch start - last-end")
for k in list(range(8, 20)) + list(range(60, 68)):
    e = ex[k]
    gap = ex[k + 1, 0] - e[2] if k + 1 < len(ex) else 0
    print("%4d %8d | %8d %8d %8d | %6d" % (k, e[0] - t0, e[1] - e[0], e[3] - e[0], e[2] - e[0], gap))
d = ex[:, 2] - ex[:, 0]
print("mean: last-start lag %.0f  wg0-end %.0f  last-end %.0f  | gap to next launch %.0f" % (
    (ex[:, 1] - ex[:, 0]).mean(), (ex[:, 3] - ex[:, 0]).mean(), d.mean(),
    (ex[1:, 0] - ex[:-1, 2]).mean()))
