"""Preconditioned conjugate gradient on the trailing block of a partially factored matrix, and the
preconditioners of the reference's PCG example (examples/PCG.{h,cpp}, examples/Preconditioner.h):
caller-side code in torch on the device; the matrix-vector product, the
pseudo-factor, the fp32 factor and the triangular solves are the library's HIP kernels.

Typical use (mixed direct / iterative, as examples/Optimizer.h:710-747): `factorUpTo(span)` leaves
the Schur complement S in the bottom-right blocks; `b` is reduced with `solveLUpTo`; then
    pcg = PCG(LowerPrecSolvePrecond(solver, data, span), TrailingOperator(solver, data, span))
    iters, res = pcg.solve(x_tail, b_tail)
solves S x = b to fp64 accuracy with a single-precision factor of S as the preconditioner."""
import numpy as np
import torch


class TrailingOperator:
    """y = S x for the symmetric block made of the lump columns from `span` on (Solver::addMvFrom);
    vectors hold the trailing rows only"""

    def __init__(self, solver, data_dev, span):
        self.solver, self.data, self.span = solver, data_dev, span
        self.n = solver.order()
        self.bar = int(solver.spanVectorOffset(span))

    def __call__(self, x_tail):
        full_in = torch.zeros(self.n, dtype=self.data.dtype, device=self.data.device)
        full_in[self.bar:] = x_tail
        full_out = torch.zeros_like(full_in)
        self.solver.addMvFrom(self.data, self.span, full_in, self.n, full_out, self.n, 1, 1.0)
        return full_out[self.bar:]


class IdentityPrecond:
    def __call__(self, r):
        return r.clone()


class BlockJacobiPrecond:
    """M^-1 r with M = the diagonal blocks of the spans (Preconditioner.h:49-106); the block
    factors come from Solver::pseudoFactorFrom on a copy of the trailing data"""

    def __init__(self, solver, data_dev, span):
        self.bar = int(solver.spanVectorOffset(span))
        sk = solver.skel()
        work = data_dev.clone()
        solver.pseudoFactorFrom(work, span)
        host = work.cpu().numpy()
        ss = sk["spanStart"]
        self.blocks = []  # (first row relative to the tail, inverse of the diagonal block)
        for s in range(span, solver.numSpans()):
            off, stride = solver.diagBlockOffsetOfSpan(s)
            n = int(ss[s + 1] - ss[s])
            L = np.tril(host[off + np.arange(n)[:, None] * stride + np.arange(n)[None, :]])
            inv = np.linalg.inv(L @ L.T)
            self.blocks.append((int(ss[s]) - self.bar, torch.from_numpy(inv).to(data_dev.device)))

    def __call__(self, r):
        out = torch.empty_like(r)
        for a, inv in self.blocks:
            n = inv.shape[0]
            out[a:a + n] = inv.to(r.dtype) @ r[a:a + n]
        return out


class LowerPrecSolvePrecond:
    """M^-1 r with M = S factored in SINGLE precision (Preconditioner.h:141-206): fp32 copy of the
    trailing data, `factorFrom(span)`, and per application solveLFrom + solveLtFrom in fp32.  The
    diagonal is perturbed (x (1 + eps) + eps, eps = 1e-8, 3e-8, ...) until the factor is finite."""

    def __init__(self, solver, data_dev, span):
        self.solver, self.span = solver, span
        self.n = solver.order()
        self.bar = int(solver.spanVectorOffset(span))
        eps = 0.0
        while True:
            self.l32 = data_dev.to(torch.float32)
            if eps > 0:
                host = self.l32.cpu().numpy()
                for s in range(span, solver.numSpans()):
                    off, stride = solver.diagBlockOffsetOfSpan(s)
                    n = int(solver.skel()["spanStart"][s + 1] - solver.skel()["spanStart"][s])
                    idx = off + (stride + 1) * np.arange(n)
                    host[idx] = host[idx] * (1.0 + eps) + eps
                self.l32 = torch.from_numpy(host).to(data_dev.device)
                eps *= 3.0
            else:
                eps = 1e-8
            solver.factorFrom(self.l32, span)
            if bool(torch.isfinite(self.l32).all()):
                break

    def __call__(self, r):
        v = torch.zeros(self.n, dtype=torch.float32, device=r.device)
        v[self.bar:] = r.to(torch.float32)
        self.solver.solveLFrom(self.l32, self.span, v, self.n, 1)
        self.solver.solveLtFrom(self.l32, self.span, v, self.n, 1)
        return v[self.bar:].to(r.dtype)


class PCG:
    """examples/PCG.cpp: plain preconditioned conjugate gradient; solve() returns
    (iterations, relative residual)"""

    def __init__(self, apply_inv_m, apply_a, wanted_residual=1e-10, max_steps=100):
        self.apply_inv_m, self.apply_a = apply_inv_m, apply_a
        self.wanted_residual, self.max_steps = wanted_residual, max_steps

    def solve(self, x, b):
        x.zero_()
        r = b.clone()
        z = self.apply_inv_m(r)
        p = z.clone()
        rz = torch.dot(r, z)
        bnorm = float(b.norm())
        res = float(r.norm()) / bnorm
        it = 0
        while it < self.max_steps and res > self.wanted_residual:
            ap = self.apply_a(p)
            alpha = rz / torch.dot(p, ap)
            x += alpha * p
            r -= alpha * ap
            res = float(r.norm()) / bnorm
            it += 1
            if res <= self.wanted_residual:
                break
            z = self.apply_inv_m(r)
            rz_new = torch.dot(r, z)
            p = z + (rz_new / rz) * p
            rz = rz_new
        return it, res
