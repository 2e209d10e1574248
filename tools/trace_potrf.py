"""In-situ clock trace of the chain potrf (needs a library built with BSP_KTRACE=1, see build.sh:
BSP_KTRACE=1 BSP_OUT=../libbaspacho_amd_trace.so BSP_BUILD_DIR=../_build_trace bash build.sh, then
BSP_LIB_PATH=baspacho_amd/libbaspacho_amd_trace.so python tools/trace_potrf.py).
Per stamped launch (workgroup 0 of the chain kernels), clocks spent in: load + pending update /
16-step loop / inverses + store; and inside step 6 of the loop: pivot chain (LDS read -> sol
written) / first barrier / thread-level update + MFMA + second barrier."""
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
tr = tr[(tr[:, 0] > 0) & (tr[:, 3] > 0)]
d = np.diff(tr[:, :4], axis=1)
print("records", len(tr), "(clock units as clock64() counts them)")
print("mean load/pre %.0f  loop %.0f  tail %.0f  total %.0f" % (*d.mean(axis=0), d.sum(axis=1).mean()))
ok = tr[:, 4] > 0
s = tr[ok]
# (blocked potrf: slots 4-7 are stamped by thread BSP_TRACE_TID inside block 1: after the publish
#  barrier / after the register factorization / after the second barrier / end of the block)
print("step 6 (blocked form: block 1): pivot chain | factor %.0f  barrier %.0f  update+mfma+barrier %.0f  | whole %.0f  (loop/16 = %.0f)" % (
    (s[:, 5] - s[:, 4]).mean(), (s[:, 6] - s[:, 5]).mean(), (s[:, 7] - s[:, 6]).mean(),
    (s[:, 7] - s[:, 4]).mean(), d[:, 1].mean() / 16))
for q in (10, 50, 90):
    print("  p%d: chain %.0f  b1 %.0f  rest %.0f" % (q, np.percentile(s[:, 5] - s[:, 4], q),
                                                   np.percentile(s[:, 6] - s[:, 5], q),
                                                   np.percentile(s[:, 7] - s[:, 6], q)))
# wall-clock check of the clock unit: total span of the trace against the factor time
print("trace span (first stamp -> last stamp): %.0f clock units" % (tr[:, 3].max() - tr[:, 0].min()))
