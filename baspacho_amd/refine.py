"""Mixed-precision solve: fp32 factor on the GPU + fp64 iterative refinement (BASELINE config 5).

The reference's analogue is the fp32-factor preconditioner of its PCG example
(examples/Preconditioner.h:141-206, LowerPrecSolvePrecond).  Caller-side code: the factor and the
triangular solves are the library's HIP kernels; the fp64 residual r = b - A x is formed with torch
index ops on the block structure (plumbing)."""
import numpy as np
import torch


class BlockSymmetricOperator:
    """y = A x for the symmetric matrix whose lower triangle is stored in skeleton layout"""

    def __init__(self, solver, data_dev):
        sk = solver.skel()
        dev = data_dev.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ccp, crs, cd = sk["chainColPtr"], sk["chainRowSpan"], sk["chainData"]
        lump_of_chain = np.repeat(np.arange(len(ccp) - 1), np.diff(ccp))
        width = (sk["lumpStart"][1:] - sk["lumpStart"][:-1])[lump_of_chain]
        rows = (sk["spanStart"][1:] - sk["spanStart"][:-1])[crs]
        nelem = t(rows * width)
        chain_id = torch.repeat_interleave(torch.arange(len(crs), device=dev), nelem)
        local = torch.arange(int(cd[-1]), device=dev) - t(cd[:-1])[chain_id]
        w = t(width)[chain_id]
        r = t(sk["spanStart"][crs])[chain_id] + local // w
        c = t(sk["lumpStart"][lump_of_chain])[chain_id] + local % w
        low = r >= c
        self.r, self.c, self.a = r[low], c[low], data_dev[low].to(torch.float64)
        self.offdiag = self.r != self.c
        self.n = solver.order()

    def __call__(self, x):
        y = torch.zeros(self.n, dtype=torch.float64, device=x.device)
        y.index_add_(0, self.r, self.a * x[self.c])
        o = self.offdiag
        y.index_add_(0, self.c[o], self.a[o] * x[self.r[o]])
        return y


def solve_refined(solver, A64_dev, b64_dev, tol=1e-10, max_iters=20):
    """Solve A x = b to fp64 accuracy with an fp32 factor.  A64_dev: fp64 matrix data (skeleton
    layout, device), b64_dev: fp64 right-hand side in the solver's internal order.
    Returns (x, iterations, relative residual history)."""
    L32 = A64_dev.to(torch.float32)
    solver.factor(L32)
    op = BlockSymmetricOperator(solver, A64_dev)
    bnorm = float(b64_dev.norm())
    x = torch.zeros_like(b64_dev)
    r = b64_dev.clone()
    hist = []
    for it in range(max_iters):
        d = r.to(torch.float32).contiguous()
        solver.solve(L32, d, solver.order(), 1)
        x += d.to(torch.float64)
        r = b64_dev - op(x)
        hist.append(float(r.norm()) / bnorm)
        if hist[-1] < tol:
            break
    return x, len(hist), hist
