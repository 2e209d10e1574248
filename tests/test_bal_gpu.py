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
