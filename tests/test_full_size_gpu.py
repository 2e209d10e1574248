"""GPU: BASELINE.json's configurations at FULL size, checked through a size-independent
property: the residual probe ||L (L^T x) - A x|| / ||A x|| for random x, with A and L applied as
block-sparse operators through the skeleton by the CPU oracle (never densified); the device
solve on that factor is compared with the oracle's solve on the same factor.  North-star tolerance: 1e-10 (fp64)."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from oracle import cref

pytestmark = pytest.mark.gpu


def _data(sol, seed, beta_factor=1.2):
    h = T.random_data(sol.dataSize(), -1.0, 1.0, seed)
    sol.damp(h, 0.0, sol.order() * beta_factor)
    return h


def _probe(sol, host_A, L_host, seed=5):
    skh = cref.SkelHandle(sol.skel())
    x = T.random_data(sol.order(), -1, 1, seed)
    return float(cref.probe_residual(skh, host_A, L_host, x))


def _check(sol, host, seeds=(5,)):
    import torch
    dev = torch.from_numpy(host).cuda()
    sol.factor(dev)
    torch.cuda.synchronize()
    L = dev.cpu().numpy()
    assert np.isfinite(L).all()
    for s in seeds:
        r = _probe(sol, host, L, s)
        assert r < 1e-10, r
    # solve on the device with the same factor, against the oracle's solve on that factor
    n = sol.order()
    b = T.random_data(n, -1, 1, 77)
    v = torch.from_numpy(b.copy()).cuda()
    sol.solve(dev, v, n, 1)
    x = v.cpu().numpy()
    ref = b.copy()
    cref.solve(sol.skel(), L, ref, n, 1)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) < 1e-10
    return L


def test_c1_block_tridiagonal_full():
    """C1: 3334 x (3x3) block-tridiagonal, automatic elimination ranges"""
    sol = B.create_solver(B.Settings(), np.full(3334, 3, dtype=np.int64), T.block_tridiagonal(3334))
    _check(sol, _data(sol, 37))


def test_c2_flat_50k_full():
    """C2: genFlat(16667, 3e-4) x 3 = 50 001 dofs, ~1 TFlop"""
    sol = B.create_solver(B.Settings(), np.full(16667, 3, dtype=np.int64), T.gen_flat(16667, 3.0e-4, 37))
    _check(sol, _data(sol, 37))


def test_c3_bal871_full():
    """C3: BAL-871-shaped Schur problem (synthetic stand-in), point elimination range given"""
    sizes, ss, _, _ = T.gen_bal_synthetic()
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 527480])
    _check(sol, _data(sol, 37), seeds=(5, 6))


def test_c4_grid_batch_full():
    """C4: 64 matrices sharing genGrid(82, 82) x 3 (seed 37 + q, damp 1.3 order), factored as ONE
    batched call; every 8th matrix is probed"""
    import torch
    sol = B.create_solver(B.Settings(), np.full(82 * 82, 3, dtype=np.int64), T.gen_grid(82, 82, 1.0, 2, 37))
    hosts = [_data(sol, 37 + q, 1.3) for q in range(64)]
    devs = [torch.from_numpy(h).cuda() for h in hosts]
    sol.factor(devs)
    torch.cuda.synchronize()
    for q in range(0, 64, 8):
        r = _probe(sol, hosts[q], devs[q].cpu().numpy(), 5 + q)
        assert r < 1e-10, (q, r)


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
