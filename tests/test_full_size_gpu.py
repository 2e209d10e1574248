"""GPU: BASELINE.json's configurations at FULL size.

Round 3: every check here can fail.  The round-2 version accepted a residual probe < 1e-10 on mock
data damped by 1.2 x order -- a factor whose ENTIRE sparse-elimination update is missing reads 6e-11
there (the diagonal is ~2e6, the off-diagonals are in (-1, 1)).  Now:
  * the device factor is compared ENTRY BY ENTRY with the CPU oracle's factor of the same matrix
    (oracle/blas_factor.c, the restatement of the reference's BLAS path: BAL-871 in ~1 s on the GPU
    box's host), on lowerMask(): relative difference < 1e-12 over everything and < 1e-11 over the
    off-diagonal entries alone (the diagonal dominates the norm of a heavily damped factor);
  * the vector probe ||L (L^T x) - A x|| / ||A x|| must be < 1e-13 (measured: ~6e-16);
  * weakly damped inputs (diagonal = 1.05 x the absolute row sum: SPD by 5 % only) and REAL
    bundle-adjustment Hessians J^T J from the device pipeline (lambda = 1e-4) are factored at full
    size -- there the off-diagonal mass is as large as the diagonal and a wrong update is an O(1) error;
  * a deliberately broken build of the plan (bsp_test_set_fault(solver, 1): the elimination update is
    never launched) must turn the C3 check red.
Protocol of the reference's FactorTest (tests/FactorTest.cpp:43-107) carried to sizes a dense LLT
cannot reach; north-star tolerance 1e-10."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import bal
from baspacho_amd import testing as T
from oracle import cref

pytestmark = pytest.mark.gpu

PROBE_TOL = 1e-13       # measured 3e-16 .. 8e-16
FACTOR_TOL = 1e-12      # || (L_gpu - L_oracle)[lowerMask] || / || L_oracle[lowerMask] ||
OFFDIAG_TOL = 1e-11     # the same over the off-diagonal entries only


def _data(sol, seed, beta_factor=1.2):
    h = T.random_data(sol.dataSize(), -1.0, 1.0, seed)
    sol.damp(h, 0.0, sol.order() * beta_factor)
    return h


def _diag_index(sol):
    sk = sol.skel()
    ls = sk["lumpStart"]
    w = (ls[1:] - ls[:-1]).astype(np.int64)
    d0 = sk["chainData"][sk["chainColPtr"][:-1]]
    rep = np.repeat(np.arange(len(w)), w)
    within = np.arange(int(w.sum())) - np.repeat(np.cumsum(w) - w, w)
    return d0[rep] + within * (w[rep] + 1)     # in matrix order: entry i = diagonal of row i


def _weak_data(sol, seed, margin=1.05):
    """uniform(-1,1) entries, diagonal = margin x absolute off-diagonal row sum: strictly diagonally
    dominant by 5 %, so SPD, with nothing for an error to hide behind"""
    h = T.random_data(sol.dataSize(), -1.0, 1.0, seed)
    rs = cref.abs_row_sums(sol.skel(), h)
    h[_diag_index(sol)] = margin * rs + 1e-3
    return h


def _probe(sol, host_A, L_host, seed=5):
    skh = cref.SkelHandle(sol.skel())
    x = T.random_data(sol.order(), -1, 1, seed)
    return float(cref.probe_residual(skh, host_A, L_host, x))


def _against_oracle(sol, host, L, tol=FACTOR_TOL, off_tol=OFFDIAG_TOL):
    ref = host.copy()
    cref.blas_factor(sol.skel(), ref, sol.sparseEliminationRanges())
    mask = sol.lowerMask()
    d = (L - ref)
    err = np.linalg.norm(d[mask]) / np.linalg.norm(ref[mask])
    assert err < tol, ("factor differs from the oracle's", err)
    mask[_diag_index(sol)] = False
    err_off = np.linalg.norm(d[mask]) / np.linalg.norm(ref[mask])
    assert err_off < off_tol, ("off-diagonal entries differ from the oracle's", err_off)
    return ref


def _check(sol, host, seeds=(5,), tol=FACTOR_TOL, off_tol=OFFDIAG_TOL, probe_tol=PROBE_TOL):
    import torch
    dev = torch.from_numpy(host).cuda()
    sol.factor(dev)
    torch.cuda.synchronize()
    L = dev.cpu().numpy()
    assert np.isfinite(L).all()
    for s in seeds:
        r = _probe(sol, host, L, s)
        assert r < probe_tol, ("residual probe", r)
    ref = _against_oracle(sol, host, L, tol, off_tol)
    # solve on the device with the same factor, against the oracle's solve on the oracle's factor
    n = sol.order()
    b = T.random_data(n, -1, 1, 77)
    v = torch.from_numpy(b.copy()).cuda()
    sol.solve(dev, v, n, 1)
    x = v.cpu().numpy()
    want = b.copy()
    cref.solve(sol.skel(), ref, want, n, 1)
    assert np.linalg.norm(x - want) / np.linalg.norm(want) < 1e-10
    return L


def test_c1_block_tridiagonal_full():
    """C1: 3334 x (3x3) block-tridiagonal, automatic elimination ranges; mock damping and 5 % dominance"""
    sol = B.create_solver(B.Settings(), np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334))
    _check(sol, _data(sol, 37))
    _check(sol, _weak_data(sol, 38))


def test_c2_flat_50k_full():
    """C2: genFlat(16667, 3e-4) x 3 = 50 001 dofs, ~1 TFlop"""
    sol = B.create_solver(B.Settings(), np.full(16667, 3, dtype=np.int64), T.gen_flat(16667, 3.0e-4, 37))
    _check(sol, _data(sol, 37))


def test_c2_flat_50k_weakly_damped():
    """C2 with the diagonal at 1.05 x the absolute row sum"""
    sol = B.create_solver(B.Settings(), np.full(16667, 3, dtype=np.int64), T.gen_flat(16667, 3.0e-4, 37))
    _check(sol, _weak_data(sol, 41), tol=1e-11, off_tol=1e-10)


@pytest.fixture(scope="module")
def bal871():
    sizes, ss, cam, pt = T.gen_bal_synthetic()
    return sizes, ss, cam, pt


def test_c3_bal871_full(bal871):
    """C3: BAL-871-shaped Schur problem (synthetic stand-in), point elimination range given"""
    sizes, ss, _, _ = bal871
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 527480])
    _check(sol, _data(sol, 37), seeds=(5, 6))


def test_c3_bal871_weakly_damped(bal871):
    """C3 with the diagonal at 1.05 x the absolute row sum: the Schur update is as large as the
    camera block it lands on"""
    sizes, ss, _, _ = bal871
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 527480])
    _check(sol, _weak_data(sol, 43), tol=1e-11, off_tol=1e-10)


def test_c3_dropped_elimination_update_is_detected(bal871, monkeypatch):
    """FAULT INJECTION: a plan whose sparse-elimination update is never launched (the failure the
    round-2 probe could not see) must fail the very checks test_c3_bal871_full applies -- the oracle
    comparison AND the probe at its new threshold"""
    import torch
    sizes, ss, _, _ = bal871
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 527480])
    sol._testSetFault(1)
    host = _data(sol, 37)
    dev = torch.from_numpy(host).cuda()
    sol.factor(dev)
    torch.cuda.synchronize()
    L = dev.cpu().numpy()
    assert np.isfinite(L).all()
    r = _probe(sol, host, L, 5)
    assert r > 10 * PROBE_TOL, ("the probe must see a factor without its Schur update", r)
    with pytest.raises(AssertionError):
        _against_oracle(sol, host, L)
    with pytest.raises(AssertionError):
        _check(sol, host, seeds=(5,))


def _hessian(sol, prob, lam):
    import torch
    pipe = bal.DevicePipeline(prob, sol)
    pipe.linearize()
    data = torch.zeros(sol.dataSize(), dtype=torch.float64, device="cuda")
    grad = torch.zeros(sol.order(), dtype=torch.float64, device="cuda")
    pipe.fill_hessian(data, grad, lam)
    torch.cuda.synchronize()
    return data, grad


def test_c3_bal871_real_hessian(bal871):
    """C3 on a REAL bundle-adjustment Hessian: a geometrically consistent 871-camera scene with the
    stand-in's co-visibility, linearised and assembled on the device (J^T J, LM damping 1e-4), factored
    and compared with the oracle.  Two correct Cholesky factors of an ill-conditioned matrix differ
    by cond x eps, so the entry-wise tolerance is looser here; the probe is backward stable and is not."""
    sizes, ss, cam, pt = bal871
    prob = bal.synth_scene_for(871, 527480, cam, pt, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 527480])
    data, grad = _hessian(sol, prob, 1e-4)
    host = data.cpu().numpy()
    assert np.isfinite(host).all()
    _check(sol, host, seeds=(5, 6), tol=1e-8, off_tol=1e-7, probe_tol=1e-12)


def test_c4_grid_batch_full():
    """C4: 64 matrices sharing genGrid(82, 82) x 3 (seed 37 + q, damp 1.3 order; every fourth one
    weakly damped instead), factored as ONE batched call; every 8th matrix and three of the weakly
    damped ones are probed and compared with the oracle"""
    import torch
    sol = B.create_solver(B.Settings(), np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37))
    hosts = [_weak_data(sol, 37 + q) if q % 4 == 3 else _data(sol, 37 + q, 1.3) for q in range(64)]
    devs = [torch.from_numpy(h).cuda() for h in hosts]
    sol.factor(devs)
    torch.cuda.synchronize()
    for q in list(range(0, 64, 8)) + [3, 31, 63]:
        L = devs[q].cpu().numpy()
        r = _probe(sol, hosts[q], L, 5 + q)
        assert r < PROBE_TOL, (q, r)
        weak = q % 4 == 3
        _against_oracle(sol, hosts[q], L, 1e-11 if weak else FACTOR_TOL, 1e-10 if weak else OFFDIAG_TOL)


def test_c4_grid_single_matrix_full():
    """C4's structure as ONE matrix -- the latency-bound schedule of round 6: the 990-column root as a
    whole in the persistent tail launch, the potrf of the small tree levels folded into their trsm
    launches -- entry by entry against the oracle, normally and weakly damped, with the solve"""
    sol = B.create_solver(B.Settings(), np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37))
    before = sol.runCounters()
    _check(sol, _data(sol, 37, 1.2), seeds=(5, 6))
    _check(sol, _weak_data(sol, 38), tol=1e-11, off_tol=1e-10)
    after = sol.runCounters()
    assert after["tail_launches"] - before["tail_launches"] == 2
    assert after["potrf_folded_levels"] - before["potrf_folded_levels"] >= 2 * 15
    assert sol.planStats()["num_tail_panels"] == 16


def test_c5_bal1723_fp32_refined_full():
    """C5: BAL-1723-shaped problem (1723 cams, 156 502 pts, ~0.68 M observations; synthetic
    stand-in), fp32 factor on the device + fp64 iterative refinement to ||r|| / ||b|| < 1e-10"""
    import torch
    from baspacho_amd.refine import solve_refined
    sizes, ss, cam, _ = T.gen_bal_synthetic(num_cams=1723, num_pts=156502, mean_track=4.95, band=24,
                                            seed=11)
    assert 0.6e6 < len(cam) < 0.76e6
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 156502])
    host = _data(sol, 37)
    A = torch.from_numpy(host).cuda()
    b = torch.from_numpy(T.random_data(sol.order(), -1, 1, 4)).cuda()
    x, iters, hist = solve_refined(sol, A, b, tol=1e-10)
    assert hist[-1] < 1e-10, hist
    assert iters <= 8, hist


@pytest.fixture(scope="module")
def bal1723_real():
    sizes, ss, cam, pt = T.gen_bal_synthetic(num_cams=1723, num_pts=156502, mean_track=4.95, band=24,
                                             seed=11)
    prob = bal.synth_scene_for(1723, 156502, cam, pt, seed=7)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 156502])
    return prob, sol


def test_c5_bal1723_real_hessian_fp32_refined(bal1723_real):
    """C5 under stress: the fp32 factor of a real bundle-adjustment Hessian with LM damping 1e-6 is a
    visibly worse inverse than that of the mock matrix above (which has cond ~ 1 and converges in
    two steps): the refinement needs three or more iterations (measured on MI355X: 2.8e-7, 6.4e-10,
    2.4e-11), must be monotone and must reach 1e-10; the refined solution agrees with the fp64 direct
    solve to the forward error the conditioning allows."""
    from baspacho_amd.refine import solve_refined
    prob, sol = bal1723_real
    A, grad = _hessian(sol, prob, 1e-6)
    x, iters, hist = solve_refined(sol, A, grad, tol=1e-10, max_iters=40)
    assert hist[-1] < 1e-10, hist
    assert iters > 2, hist
    assert all(b < a for a, b in zip(hist, hist[1:])), hist
    L = A.clone()
    sol.factor(L)
    want = grad.clone()
    sol.solve(L, want, sol.order(), 1)
    # (forward error = cond x residual: measured 1e-5 at a residual of 2e-11, i.e. cond ~ 5e5)
    assert float((x - want).norm() / want.norm()) < 1e-4


def test_c5_bal1723_real_hessian_fp32_preconditioned_cg(bal1723_real):
    """... and with LM damping 1e-8 plain refinement STALLS (the gauge modes: 2.9e-7, 6.5e-9, 6.5e-9,
    ... measured): this is where the reference uses its fp32 factor as a PRECONDITIONER of conjugate
    gradients (examples/Preconditioner.h:141-206, examples/PCG.cpp) -- fp32 factor + solves, fp64
    products A p through addMvFrom, all library kernels.  Must reach 1e-10."""
    import torch
    from baspacho_amd.pcg import PCG, LowerPrecSolvePrecond, TrailingOperator
    prob, sol = bal1723_real
    A, grad = _hessian(sol, prob, 1e-8)
    pcg = PCG(LowerPrecSolvePrecond(sol, A, 0), TrailingOperator(sol, A, 0), wanted_residual=1e-10,
              max_steps=60)
    x = torch.zeros_like(grad)
    iters, res = pcg.solve(x, grad)
    assert res < 1e-10, (iters, res)
    assert iters >= 2, iters
    r = grad.clone()
    sol.addMvFrom(A, 0, x, sol.order(), r, sol.order(), 1, -1.0)
    assert float(r.norm() / grad.norm()) < 1e-9


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_data_offsets_beyond_32_bits(prec):
    """maximum sizes: ONE dense lump of order 65 600 = 4.3e9 values (34 GB in fp64), so that element
    offsets pass 2^31 AND 2^32: factor (vector probe), solveL, solveLt, solve -- tools/huge_lump.py,
    which builds and checks everything on the device in row blocks (a subprocess: 70 GB of device
    memory are released when it ends)"""
    import os
    import subprocess
    import sys
    import torch
    need = 90e9 if prec == "f64" else 50e9
    if torch.cuda.mem_get_info()[0] < need:
        pytest.skip("needs %.0f GB of free device memory" % (need / 1e9))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "huge_lump.py"), "65600", prec],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "OK" in r.stdout
