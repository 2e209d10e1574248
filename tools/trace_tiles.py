"""In-situ trace of a chain-step launch (library built with BSP_KTRACE=1 -DBSP_TRACE_TILE): per
launch, when workgroup 0 (next potrf) and the LAST tile workgroup start and end, against each other.
Records: workgroup 0 -> slots 0-3, last tile workgroup -> slots 4-7 (separate records)."""
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
for it in range(3):
    buf = A.clone()
    torch.cuda.synchronize()
    bsp.debug_read_trace()
    sol.factor(buf)
    torch.cuda.synchronize()
    tr = bsp.debug_read_trace()
w0 = tr[tr[:, 0] > 0]
tl = tr[(tr[:, 0] == 0) & (tr[:, 4] > 0)]
w0 = w0[np.argsort(w0[:, 0])]
tl = tl[np.argsort(tl[:, 4])]
print("workgroup-0 records %d, tile records %d" % (len(w0), len(tl)))
# pair every tile record with the workgroup-0 record whose start is nearest before its end
t0 = w0[:, 0].min()
print(" launch   wg0: start   end  | last tile: start  solve  mult   end | tile_end - wg0_end")
k = 0
rows = []
for r in tl:
    while k + 1 < len(w0) and w0[k + 1, 0] <= r[4] + 20:
        k += 1
    a = w0[k]
    rows.append((a[0] - t0, a[3] - t0, r[4] - t0, r[5] - r[4], r[6] - r[5], r[7] - r[6], r[7] - a[3], a[3] - a[0], r[7] - r[4], r[4] - a[0]))
rows = np.array(rows, dtype=np.float64)
for q in range(0, len(rows), 8):
    x = rows[q]
    print("%6d %12d %6d | %12d %6d %6d %6d | %8d" % (q, x[0], x[7], x[2] - x[0], x[3], x[4], x[5], x[6]))
print("(clock = wall_clock64, 100 MHz: 1 unit = 10 ns)")
print("mean wg0 duration %.0f  tile duration %.0f (solve %.0f mult %.0f store %.0f)  tile start lag %.0f  tile_end - wg0_end %.0f" % (
    rows[:, 7].mean(), rows[:, 8].mean(), rows[:, 3].mean(), rows[:, 4].mean(), rows[:, 5].mean(), rows[:, 9].mean(), rows[:, 6].mean()))
