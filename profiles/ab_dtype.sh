# factor() time of a workload in fp64 and fp32 under schedule switches
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python - <<PY
import sys, time, torch
sys.path.insert(0, ".")
import bench
import baspacho_amd as B
from baspacho_amd import testing as T
dev = torch.device("cuda", 0)
out = []
for wl in ("bal871", "bal1723"):
    sizes, ss, ranges, desc, _ = bench.build_problem(wl)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    sol.setStream(torch.cuda.current_stream(dev))
    h = T.random_data(sol.dataSize(), -1.0, 1.0, 37)
    sol.damp(h, 0.0, sol.order() * 1.2)
    A = torch.from_numpy(h).to(dev)
    for dt in (torch.float64, torch.float32):
        bufs = [A.to(dt).clone() for _ in range(6)]
        sol.factor(bufs[0]); sol.factor(bufs[1]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, 6):
            sol.factor(bufs[i])
        torch.cuda.synchronize()
        out.append("%s/%s %.3f ms" % (wl, "f64" if dt == torch.float64 else "f32", (time.perf_counter() - t0) / 4 * 1e3))
        del bufs
print("%-40s %s" % ("$cfg", "   ".join(out)))
PY
done
