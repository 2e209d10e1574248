"""one-off: refinement history of the fp32 factor on the real BAL-1723-shaped Hessian for several LM dampings"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import baspacho_amd as B
from baspacho_amd import bal, testing as T
from baspacho_amd.refine import solve_refined
from test_full_size_gpu import _hessian
sizes, ss, cam, pt = T.gen_bal_synthetic(num_cams=1723, num_pts=156502, mean_track=4.95, band=24, seed=11)
sol = B.create_solver(B.Settings(), sizes, ss, [0, 156502])
for noise, perturb in ((0.5, 1e-2), (0.5, 0.1)):
    prob = bal.synth_scene_for(1723, 156502, cam, pt, seed=7, noise=noise, perturb=perturb)
    for lam in (1e-4, 1e-6, 1e-8, 1e-10, 0.0):
        A, grad = _hessian(sol, prob, lam)
        try:
            x, iters, hist = solve_refined(sol, A, grad, tol=1e-10, max_iters=40)
            print(noise, perturb, lam, iters, ["%.1e" % h for h in hist[:8]])
        except Exception as e:
            print(lam, "failed", e)
