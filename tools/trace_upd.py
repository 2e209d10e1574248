"""In-situ trace of one workgroup per updateTile launch (library built with
BSP_KTRACE=1 BSP_EXTRA_DEFS=-DBSP_TRACE_UPD, see hip_kernels.h UPD_STAMP):
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace_upd.so python tools/trace_upd.py [batch]
Per launch: workgroups, K, and the traced workgroup's phases in us (wall clock, 100 MHz)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import baspacho_amd as bsp
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
sizes, ss, ranges, desc, src = bench.build_problem("grid82", None)
sol = bsp.create_solver(bsp.Settings(), sizes, ss, ranges)
sol.setStream(torch.cuda.current_stream(dev))
from baspacho_amd import testing as T
hs = []
for q in range(batch):
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37 + q)
    sol.damp(h, 0.0, sol.order() * 1.3)
    hs.append(torch.from_numpy(h).to(dev))
for it in range(3):
    bufs = [a.clone() for a in hs]
    torch.cuda.synchronize()
    bsp.debug_read_trace()
    sol.factor(bufs if batch > 1 else bufs[0])
    torch.cuda.synchronize()
    tr = bsp.debug_read_trace()
tr = tr[tr[:, 7] < 0]
print("%6s %5s | %7s %7s %7s %7s %7s | %7s" % ("wgs", "K", "desc", "chunk0", "K loop", "old", "store", "total"))
tot = np.zeros(6)
for r in tr:
    t = r[:6].astype(float) * 0.01
    old = t[4] - t[3] if r[4] > 0 else 0.0
    end_from = t[4] if r[4] > 0 else t[3]
    row = [t[1] - t[0], t[2] - t[1], t[3] - t[2], old, t[5] - end_from, t[5] - t[0]]
    tot += row
    print("%6d %5d | %7.2f %7.2f %7.2f %7.2f %7.2f | %7.2f" % (-r[7], r[6], *row))
print("sum over %d launches: desc %.1f chunk0 %.1f loop %.1f old %.1f store %.1f total %.1f us" % (len(tr), *tot))
