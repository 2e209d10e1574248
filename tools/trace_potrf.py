"""In-situ clock trace of the chain potrf (needs a library built with BSP_KTRACE=1):
per launch, clocks spent in load / 16-step loop / store, for a factor with and without the
lookahead side stream."""
import os, sys
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
d = np.diff(tr, axis=1) / 2400.0  # us at 2.4 GHz (s_memtime may tick at 100 MHz: check scale)
print("records", len(tr))
print("idx   load    loop   store   total")
for i in range(0, min(len(d), 48)):
    print("%3d %7.1f %7.1f %7.1f %7.1f" % (i, d[i, 0], d[i, 1], d[i, 2], d[i].sum()))
print("mean", d.mean(axis=0), d.sum(axis=1).mean())
