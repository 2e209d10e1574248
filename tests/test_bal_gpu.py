"""GPU tests of the BAL caller pipeline (SURVEY.md 8 row f3, and a9's deviceAccessor()): a small
synthetic BAL file -> loader -> solver -> device linearisation -> Hessian assembly through
Solver::deviceAccessor() inside a HIP kernel -> factor + solve of the damped normal equations."""
import numpy as np
import pytest
import torch

import baspacho_amd as B
from baspacho_amd import bal
from oracle import bal_model, cref

pytestmark = pytest.mark.gpu


def _setup(tmp_path, num_cams, num_pts, seed):
    prob0 = bal.synth_scene(num_cams=num_cams, num_pts=num_pts, seed=seed)
    path = tmp_path / ("problem-%d-%d-pre.txt" % (num_cams, num_pts))
    bal.save_bal(path, prob0)
    prob = bal.load_bal(path)
    sizes, ss, ranges = bal.bal_structure(prob)
    sol = B.create_solver(B.Settings(), sizes, ss, ranges)
    return prob, sol


def test_device_accessor_offsets_are_the_host_accessor(tmp_path):
    """PermutedCoalescedAccessor::blockOffset / diagBlockOffset evaluated INSIDE the fill kernel
    (Accessor.h:145-166 on the device) == the host accessor, bit-exactly (integer work)"""
    prob, sol = _setup(tmp_path, 14, 260, 7)
    pipe = bal.DevicePipeline(prob, sol)
    pipe.linearize()
    data = torch.zeros(sol.dataSize(), dtype=torch.float64, device="cuda")
    dbg = torch.zeros(7 * pipe.n_obs, dtype=torch.int64, device="cuda")
    pipe.fill_hessian(data, None, 0.0, dbg)
    res, Jc, Jp = bal_model.linearize(prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy)
    _, _, offs = bal_model.fill_hessian_host(sol, prob, Jc, Jp, res, 0.0)
    assert np.array_equal(dbg.cpu().numpy().reshape(-1, 7), offs)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bal_file_to_factored_step(tmp_path, dtype):
    prob, sol = _setup(tmp_path, 16, 400, 11)
    pipe = bal.DevicePipeline(prob, sol)
    # ---- linearisation on the device vs the numpy dual-number oracle
    pipe.linearize()
    res, Jc, Jp = bal_model.linearize(prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy)
    assert np.allclose(pipe.res.cpu().numpy().reshape(-1, 2), res, rtol=1e-11, atol=1e-9)
    assert np.allclose(pipe.Jc.cpu().numpy().reshape(-1, 2, 9), Jc, rtol=1e-10, atol=1e-8)
    assert np.allclose(pipe.Jp.cpu().numpy().reshape(-1, 2, 3), Jp, rtol=1e-10, atol=1e-8)
    # ---- Hessian / gradient assembly through the device accessor vs the host accessor
    lam = 1e-2
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    data = torch.zeros(sol.dataSize(), dtype=tdt, device="cuda")
    grad = torch.zeros(sol.order(), dtype=tdt, device="cuda")
    pipe.fill_hessian(data, grad, lam)
    hdata, hgrad, _ = bal_model.fill_hessian_host(sol, prob, Jc, Jp, res, lam)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    mask = sol.lowerMask()
    got = data.cpu().numpy().astype(np.float64)
    assert np.linalg.norm((got - hdata)[mask]) <= tol * np.linalg.norm(hdata[mask])
    assert np.linalg.norm(grad.cpu().numpy() - hgrad) <= tol * 10 * np.linalg.norm(hgrad)
    # ---- factor + solve of the damped normal equations on the device vs the oracle
    step = grad.clone()
    sol.factor(data)
    sol.solve(data, step, sol.order(), 1)
    ref = hdata.copy()
    cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
    lo = data.cpu().numpy().astype(np.float64)
    ftol = 1e-10 if dtype == np.float64 else 5e-4
    assert np.linalg.norm((lo - ref)[mask]) / np.linalg.norm(ref[mask]) < ftol
    want = hgrad.copy()
    cref.solve(sol.skel(), ref, want, sol.order(), 1)
    assert np.linalg.norm(step.cpu().numpy() - want) / np.linalg.norm(want) < (1e-9 if dtype == np.float64 else 5e-3)
    if dtype == np.float64:
        # the LM step reduces the cost (vector in internal order: spanVectorOffset(paramToSpan[i]))
        s = step.cpu().numpy()
        newp = bal.BalProblem(prob.cams.copy(), prob.pts.copy(), prob.obs_cam, prob.obs_pt, prob.obs_xy)
        perm = sol.paramToSpan()
        for i in range(prob.num_pts):
            o = sol.spanVectorOffset(int(perm[i]))
            newp.pts[i] -= s[o:o + 3]
        for i in range(prob.num_cams):
            o = sol.spanVectorOffset(int(perm[prob.num_pts + i]))
            newp.cams[i] -= s[o:o + 9]
        cost0 = 0.5 * np.sum(res ** 2)
        res1, _, _ = bal_model.linearize(newp.cams[newp.obs_cam], newp.pts[newp.obs_pt], newp.obs_xy)
        assert 0.5 * np.sum(res1 ** 2) < 0.5 * cost0


def _setup_se3(tmp_path, num_cams, num_pts, seed, perturb=1e-2):
    prob0 = bal.synth_scene(num_cams=num_cams, num_pts=num_pts, seed=seed, perturb=perturb)
    path = tmp_path / ("problem-%d-%d-pre.txt" % (num_cams, num_pts))
    bal.save_bal(path, prob0)
    prob = bal.load_bal(path)
    sizes, ss, ranges = bal.bal_structure(prob, cam_size=6)
    return prob, B.create_solver(B.Settings(), sizes, ss, ranges)


def test_se3_parameterisation_matches_the_reference_model(tmp_path):
    """the reference optimizer's parameterisation (BaAtLarge.h:56-150, BaAtLargeOptimizer.cpp:24-52,
    100-131): 6-wide camera blocks = SE3 tangent of a left perturbation, calibration fixed.  Device
    linearisation == oracle (pinned by finite differences on the CPU), device Hessian / gradient
    through deviceAccessor() == host-accessor assembly, LM step == the oracle's step."""
    prob, sol = _setup_se3(tmp_path, 16, 400, 11)
    pipe = bal.DevicePipeline(prob, sol, param="se3")
    pipe.linearize()
    res, Jc, Jp = bal_model.linearize_se3(prob.cams[prob.obs_cam], prob.pts[prob.obs_pt], prob.obs_xy)
    assert np.allclose(pipe.res.cpu().numpy().reshape(-1, 2), res, rtol=1e-11, atol=1e-9)
    assert np.allclose(pipe.Jc.cpu().numpy().reshape(-1, 2, 6), Jc, rtol=1e-10, atol=1e-8)
    assert np.allclose(pipe.Jp.cpu().numpy().reshape(-1, 2, 3), Jp, rtol=1e-10, atol=1e-8)
    lam = 1e-5
    data = torch.zeros(sol.dataSize(), dtype=torch.float64, device="cuda")
    grad = torch.zeros(sol.order(), dtype=torch.float64, device="cuda")
    pipe.fill_hessian(data, grad, lam)
    hdata, hgrad = bal_model.fill_hessian_host_se3(sol, prob, Jc, Jp, res, lam)
    mask = sol.lowerMask()
    got = data.cpu().numpy()
    assert np.linalg.norm((got - hdata)[mask]) <= 1e-13 * np.linalg.norm(hdata[mask])
    assert np.linalg.norm(grad.cpu().numpy() - hgrad) <= 1e-12 * np.linalg.norm(hgrad)
    step = grad.clone()
    sol.factor(data)
    sol.solve(data, step, sol.order(), 1)
    ref = hdata.copy()
    cref.factor(sol.skel(), ref, sol.sparseEliminationRanges())
    want = hgrad.copy()
    cref.solve(sol.skel(), ref, want, sol.order(), 1)
    assert np.linalg.norm(step.cpu().numpy() - want) / np.linalg.norm(want) < 1e-7
    # a solver with 9-wide camera blocks is refused (the kernels would write out of bounds)
    sizes9, ss9, ranges9 = bal.bal_structure(prob, cam_size=9)
    sol9 = B.create_solver(B.Settings(), sizes9, ss9, ranges9)
    pipe9 = bal.DevicePipeline(prob, sol9, param="se3")
    pipe9.linearize()
    with pytest.raises(RuntimeError):
        pipe9.fill_hessian(torch.zeros(sol9.dataSize(), dtype=torch.float64, device="cuda"), None, 0.0)


def test_lm_loop_of_the_reference_optimizer_converges(tmp_path):
    """BAL_opt's Levenberg-Marquardt loop (BaAtLargeOptimizer.cpp:186-234: lambda schedule, exp-map
    update T <- exp(-step) T, acceptance rule) with every numeric stage on the device: from a
    perturbed scene the cost falls by orders of magnitude to the noise floor"""
    prob, sol = _setup_se3(tmp_path, 14, 300, 3, perturb=2e-2)
    cost0 = bal.total_cost(prob)
    out, hist = bal.lm_optimize(prob, sol, max_iters=25)
    cost1 = bal.total_cost(out)
    assert cost1 < 0.02 * cost0, (cost0, cost1, hist)
    # noise floor: 0.5 px of Gaussian noise on 2 coordinates per observation
    assert cost1 < 2.0 * 0.5 * 0.25 * 2 * len(prob.obs_cam), (cost1, len(prob.obs_cam))
    accepted = [h for h in hist if h[1] <= h[0]]
    assert len(accepted) >= 3
