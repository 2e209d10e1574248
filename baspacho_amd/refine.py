"""Mixed-precision solve: fp32 factor on the GPU + fp64 iterative refinement (BASELINE config 5).

The reference's analogue is the fp32-factor preconditioner of its PCG example
(examples/Preconditioner.h:141-206, LowerPrecSolvePrecond).  The three heavy steps are library
kernels -- factor<float>, solve<float>, and the fp64 residual r = b - A x through Solver::addMvFrom
(Solver.h:89-91) on the un-factored fp64 matrix.  The glue around them is torch: the fp64 <-> fp32
casts of the correction, the update x += d, the copy of b into r before addMvFrom, and the two norms
of the stopping test (O(order) element-wise work per iteration, against O(factor size) in the solve
and the residual)."""
import torch


def solve_refined(solver, A64_dev, b64_dev, tol=1e-10, max_iters=20, timings=None):
    """Solve A x = b to fp64 accuracy with an fp32 factor.  A64_dev: fp64 matrix data (skeleton
    layout, device, left untouched), b64_dev: fp64 right-hand side in the solver's internal
    order.  Returns (x, iterations, relative residual history).  timings (dict, optional) receives
    factor_f32_ms / refine_ms measured with torch.cuda events."""
    n = solver.order()
    L32 = A64_dev.to(torch.float32)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timings is not None else None
    if ev:
        ev[0].record()
    solver.factor(L32)
    if ev:
        ev[1].record()
    bnorm = float(b64_dev.norm())
    x = torch.zeros_like(b64_dev)
    r = b64_dev.clone()
    hist = []
    for it in range(max_iters):
        d = r.to(torch.float32).contiguous()
        solver.solve(L32, d, n, 1)
        x += d.to(torch.float64)
        r = b64_dev.clone()
        solver.addMvFrom(A64_dev, 0, x, n, r, n, 1, -1.0)   # r = b - A x, fp64, library kernel
        hist.append(float(r.norm()) / bnorm)
        if hist[-1] < tol:
            break
    if ev:
        ev[2].record()
        torch.cuda.synchronize()
        timings["factor_f32_ms"] = ev[0].elapsed_time(ev[1])
        timings["refine_ms"] = ev[1].elapsed_time(ev[2])
    return x, len(hist), hist
