import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import baspacho_amd as B
from baspacho_amd import testing as T
def run(name, sizes, ss, ranges=()):
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream())
    h = T.random_data(sol.dataSize(), -1, 1, 37); sol.damp(h, 0.0, sol.order()*1.2)
    A = torch.from_numpy(h).cuda()
    bufs = [A.clone() for _ in range(12)]
    sol.factor(bufs[0]); sol.factor(bufs[1]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 12): sol.factor(bufs[i])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = sol.planStats()
    print("%-10s host enqueue %.3f ms/factor, total %.3f ms/factor, launches %d" % (name, (t1-t0)*100, (t2-t0)*100, st["num_launches"]))
run("grid82", np.full(82*82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37))
run("tridiag", np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334))
run("flat50k", np.full(16667, 3, dtype=np.int64), T.gen_flat(16667, 3.0e-4, 37))
s, ss, _, _ = T.gen_bal_synthetic(num_cams=120, num_pts=40000, band=16)
run("bal-small", s, ss, [0, 40000])
