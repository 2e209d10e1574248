"""a factor whose data offsets pass 2^31 (and 2^32) elements: ONE dense lump of order n (default
65 600: 4.3e9 values, 34 GB in fp64), built from an explicit skeleton, data generated and checked on
the device: vector probe ||L (L^T x) - A x|| / ||A x||, solveL, solveLt, solve.  The reference products
run in row blocks of 2048 (one torch matrix-vector product over more than 2^32 elements returned
wrong values in fp64 on this stack).  python tools/huge_lump.py [n] [f32|f64]"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import baspacho_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65600
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
assert n % 8 == 0
dev = torch.device("cuda", 0)
nspan = n // 8
span_start = np.arange(nspan + 1, dtype=np.int64) * 8
sol = B.Solver.from_skeleton(span_start, [0, nspan], [0, nspan], list(range(nspan)))
assert sol.order() == n and sol.dataSize() == n * n, (sol.order(), sol.dataSize())
print("order", n, "data", sol.dataSize(), "values = %.2f x 2^31" % (sol.dataSize() / 2.0**31), dt)
RB = 2048


def mv_lower(M, x):
    """tril(M) @ x"""
    y = torch.zeros_like(x)
    for a in range(0, n, RB):
        b = min(n, a + RB)
        y[a:b] = M[a:b, :a] @ x[:a] + torch.tril(M[a:b, a:b]) @ x[a:b]
    return y


def mv_lower_t(M, x, strict=False):
    """tril(M)^T @ x (strict: without the diagonal)"""
    y = torch.zeros_like(x)
    for a in range(0, n, RB):
        b = min(n, a + RB)
        y[:a] += M[a:b, :a].T @ x[a:b]
        y[a:b] += torch.tril(M[a:b, a:b], -1 if strict else 0).T @ x[a:b]
    return y


g = torch.Generator(device=dev)
g.manual_seed(7)
A0 = torch.rand(n * n, dtype=dt, device=dev, generator=g).mul_(2).sub_(1)
A0.view(n, n).diagonal().add_(1.5 * n)
data = A0.clone()
sol.setStream(torch.cuda.current_stream(dev))
torch.cuda.synchronize()
t0 = time.time()
sol.factor(data)
torch.cuda.synchronize()
t1 = time.time()
print("factor %.3f s = %.1f TF/s (first call: includes the plan upload)" % (t1 - t0, n**3 / 3.0 / (t1 - t0) / 1e12))
x = torch.randn(n, dtype=dt, device=dev, generator=g)
Lm, Am = data.view(n, n), A0.view(n, n)
Lt_x = mv_lower_t(Lm, x)
y1 = mv_lower(Lm, Lt_x)
y2 = mv_lower(Am, x) + mv_lower_t(Am, x, strict=True)
rel = lambda u, v: float(torch.linalg.norm(u - v) / torch.linalg.norm(v))
probe = rel(y1, y2)
print("probe %.3e" % probe)
bL = y2.clone()
sol.solveL(data, bL, n, 1)
eL = rel(bL, Lt_x)
print("solveL error %.3e" % eL)
bLt = Lt_x.clone()
sol.solveLt(data, bLt, n, 1)
eLt = rel(bLt, x)
print("solveLt error %.3e" % eLt)
b = y2.clone()
sol.solve(data, b, n, 1)
serr = rel(b, x)
print("solve error %.3e" % serr)
tol = 1e-12 if dt == torch.float64 else 1e-4
ok = probe < tol and max(eL, eLt, serr) < tol * 100 and bool(torch.isfinite(data).all())
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
