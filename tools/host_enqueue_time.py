"""host-side enqueue time of factor() against its GPU time: python tools/host_enqueue_time.py WORKLOAD...
(a factor() call returns when everything is enqueued; when the enqueue takes as long as the
kernels, the host is the bottleneck of a launch-bound structure)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import baspacho_amd as B
from baspacho_amd import testing as T

for name in sys.argv[1:]:
    sizes, ss, ranges, desc, _ = bench.build_problem(name)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream())
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).cuda()
    bufs = [A.clone() for _ in range(8)]
    torch.cuda.synchronize()
    enq, tot = [], []
    for b in bufs:
        t0 = time.perf_counter()
        sol.factor(b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
    st = sol.planStats()
    print("%-10s launches %4d  enqueue %.3f ms (%.1f us/launch)  until done %.3f ms" % (
        name, st["num_launches"], np.median(enq[2:]), 1e3 * np.median(enq[2:]) / st["num_launches"],
        np.median(tot[2:])))
