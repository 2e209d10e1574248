"""Backward pass over a sparse-elimination range with the right-hand sides across the lanes
(csrc/hip_solve_wide.h, K-S3t + K-S3w): parity with the oracle's solve (sparseElim_subDiagMultT +
sparseElim_diagSolveLt, MatOpsCuda.cu:949-1012 restated in oracle/) on the GPU's own factor, for every
lump width the kernel is instantiated for, blocks taller than one pass, right-hand-side counts around
the group size of 16, batches, both precisions, and the paths around it (switched off, mixed widths)."""
import numpy as np
import pytest

import baspacho_amd as B
from baspacho_amd import testing as T
from oracle import cref
from helpers import spd_data, to_dev

pytestmark = pytest.mark.gpu


def _bipartite(num_pts, num_cams, pt_size, cam_size, seed, track=4):
    """points (eliminated, pt_size wide) x cameras (cam_size rows per block), `track` cameras per point"""
    rng = T.Rng(seed)
    pt = np.repeat(np.arange(num_pts, dtype=np.int64), track)
    cam = np.floor(rng.unit(num_pts * track) * num_cams).astype(np.int64)
    key = np.unique(pt * num_cams + cam)
    pt, cam = key // num_cams, key % num_cams
    sizes = np.concatenate([np.asarray(pt_size if np.ndim(pt_size) else np.full(num_pts, pt_size), dtype=np.int64),
                            np.full(num_cams, cam_size, dtype=np.int64)])
    ss = T.structure_from_pairs(num_pts + num_cams, num_pts + cam, pt)
    return sizes, ss


def _solve_both(sol, data, nrhs, dtype, seed=5):
    n = sol.order()
    d = to_dev(data.astype(dtype))
    sol.factor(d)
    rhs = T.random_data(n * nrhs, -1, 1, seed)
    v = to_dev(rhs.astype(dtype))
    c0 = sol.runCounters()
    sol.solve(d, v, n, nrhs)
    c1 = sol.runCounters()
    got = v.cpu().numpy().astype(np.float64)
    want = rhs.copy()
    cref.solve(sol.skel(), d.cpu().numpy().astype(np.float64), want, n, nrhs)
    return got, want, c1["solve_wide_launches"] - c0["solve_wide_launches"]


@pytest.mark.parametrize("nrhs", [2, 5, 16, 17, 35])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bundle_adjustment_shape(dtype, nrhs):
    """3-wide points under 9-row camera blocks: the 27 words of a block are one pass"""
    sizes, ss, _, _ = T.gen_bal_synthetic(num_cams=40, num_pts=3000, band=8, seed=5)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 3000])
    data = spd_data(sol, 11, beta_factor=1.2)
    got, want, launches = _solve_both(sol, data, nrhs, dtype)
    assert launches == 1
    tol = 1e-11 if dtype == np.float64 else 2e-4
    assert np.linalg.norm(got - want) <= tol * np.linalg.norm(want)


@pytest.mark.parametrize("pt_size,cam_size", [(1, 5), (1, 40), (2, 7), (2, 37), (3, 25), (4, 6), (4, 19)])
def test_every_width_and_blocks_taller_than_a_pass(pt_size, cam_size):
    """the kernel takes 16 / 16 / 10 / 8 rows of a block per pass for lumps 1 / 2 / 3 / 4 wide: blocks
    below and above that, a lump count that is not a multiple of the 16 per workgroup"""
    sizes, ss = _bipartite(1003, 23, pt_size, cam_size, seed=9 + pt_size)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 1003])
    data = spd_data(sol, 3, beta_factor=1.2)
    got, want, launches = _solve_both(sol, data, 7, np.float64)
    assert launches == 1
    assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)


def test_mixed_widths_take_the_row_per_lane_kernel():
    """a range whose lumps differ in width has no common instantiation: K-S3m runs, same answer"""
    pt_size = 1 + (np.arange(600) % 3)
    sizes, ss = _bipartite(600, 17, pt_size, 8, seed=4)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 600])
    data = spd_data(sol, 3, beta_factor=1.2)
    got, want, launches = _solve_both(sol, data, 6, np.float64)
    assert launches == 0
    assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)


def test_switched_off_and_one_right_hand_side():
    sizes, ss = _bipartite(500, 11, 3, 9, seed=2)
    st = B.Settings()
    st.hipOptions = {"solve_wide": 0}
    sol = B.create_solver(st, sizes, ss, [0, 500])
    data = spd_data(sol, 3, beta_factor=1.2)
    got, want, launches = _solve_both(sol, data, 6, np.float64)
    assert launches == 0
    assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)
    sol2 = B.create_solver(B.Settings(), sizes, ss, [0, 500])
    got, want, launches = _solve_both(sol2, data, 1, np.float64)
    assert launches == 0  # one right-hand side: K-S3
    assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_and_backward_only(dtype):
    """solveLt alone (the rows below the range are taken as they are) and a batch of matrices, each with
    its own [row][16] copy"""
    import torch
    sizes, ss = _bipartite(700, 13, 3, 9, seed=6)
    sol = B.create_solver(B.Settings(), sizes, ss, [0, 700])
    n, nrhs, batch = sol.order(), 19, 3
    mats, vecs, wants = [], [], []
    for q in range(batch):
        data = spd_data(sol, 20 + q, beta_factor=1.2).astype(dtype)
        d = to_dev(data)
        sol.factor(d)
        rhs = T.random_data(n * nrhs, -1, 1, 30 + q)
        want = rhs.copy()
        cref.solve_lt(sol.skel(), d.cpu().numpy().astype(np.float64), want, n, nrhs)
        mats.append(d)
        vecs.append(to_dev(rhs.astype(dtype)))
        wants.append(want)
    c0 = sol.runCounters()
    sol.solveLt(mats, vecs, n, nrhs)
    assert sol.runCounters()["solve_wide_launches"] - c0["solve_wide_launches"] == 1
    tol = 1e-11 if dtype == np.float64 else 2e-4
    for q in range(batch):
        got = vecs[q].cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - wants[q]) <= tol * np.linalg.norm(wants[q]), q
    torch.cuda.synchronize()
